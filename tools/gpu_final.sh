#!/bin/bash
# last visit of a round: GPU suite, config-2 bench line, per-step kernel table (in that order; a short budget may cut the tail)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT; TAG=${1:-r02}
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; grep -n "passed\|failed" $OUT/pytest_gpu.log
( timeout 200 python bench.py 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config2.json; cut -c1-220 $OUT/${TAG}_bench_config2.json
bash tools/prof.sh ${TAG}c2 26 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
