#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity_full.py -m gpu -q -s -rA -k "fp32x3 or known_answers or prefetcher" > $OUT/parity_full2.log 2>&1
grep -n "^\[\|passed\|failed\|^E " $OUT/parity_full2.log | head -60
bash tools/prof.sh r2base 10 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
head -70 $OUT/r2base_stats.md
for c in 3 4 5; do ( timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $OUT/bench_c$c.log; cut -c1-400 $OUT/bench_c$c.log; done
