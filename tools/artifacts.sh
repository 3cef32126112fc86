#!/bin/bash
# Round artifacts on one box: bench lines (all configs), rocprofv3 --kernel-trace --stats CSV of the bench command, PMC passes of the
# dominant GEMM, per-kernel summary with register / LDS columns.  Everything lands in gpurun_out/; copy what is judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r02}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
( timeout 900 python bench.py 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config2.json
for c in 3 4 5; do ( timeout 600 python bench.py --config $c --steps 30 --warmup 5 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config$c.json; done
cd /tmp; rm -rf /tmp/st
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-padded-compare > $OUT/${TAG}_stats_run.log 2>&1
cp $(find /tmp/st -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_rocprofv3_kernel_stats.csv
cd $R
bash tools/prof.sh ${TAG}c2 26 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
bash tools/prof.sh ${TAG}c4 13 python $R/bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
bash tools/pmc_nt256.sh > $OUT/${TAG}_pmc_nt256.log 2>&1
cp $OUT/pmc_nt256.json $OUT/${TAG}_pmc_nt256.json
cut -c1-400 $OUT/${TAG}_bench_config2.json; head -12 $OUT/${TAG}_rocprofv3_kernel_stats.csv | cut -c1-160; tail -25 $OUT/${TAG}_pmc_nt256.log
