#!/bin/bash
# round-3 visit B: new kernels' tests, then whole suite, then A/B of the deferred hybrid weight-gradient launch and of the NT head+tail plan
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp; TAG=${1:-r03b}
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "hybrid or wgrad" > $OUT/${TAG}_pytest_hybrid.log 2>&1; echo "hybrid rc=$?"; tail -5 $OUT/${TAG}_pytest_hybrid.log
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|Error\|^\[config\|^\[2-rank\|^\[conv" $OUT/${TAG}_pytest_gpu.log | head -40
bash tools/ab_env.sh "UVTG_TN_DEFER_OFF=1" "" 2>&1 | tee $OUT/${TAG}_ab_defer.log
bash tools/ab_env.sh "UVTG_NT_SPLIT_OFF=1" "" --variant A 2>&1 | tee $OUT/${TAG}_ab_split_variantA.log
bash tools/ab_env.sh "UVTG_NT_SPLIT_OFF=1" "" --config 3 2>&1 | tee $OUT/${TAG}_ab_split_config3.log
