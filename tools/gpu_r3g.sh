#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp; TAG=r03h
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "hybrid or wgrad" > $OUT/${TAG}_pytest_hybrid.log 2>&1; echo "hybrid rc=$?"; tail -5 $OUT/${TAG}_pytest_hybrid.log
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|Error" $OUT/${TAG}_pytest_gpu.log | head -20
bash tools/ab_env.sh "UVTG_TN_TAILDEFER=1" "" 2>&1 | tee $OUT/${TAG}_ab_taildefer.log
bash tools/prof.sh ${TAG}c2 26 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
head -40 $OUT/${TAG}c2_stats.md | cut -c1-150 | grep -v "at6native"
