#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp; TAG=r03i
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "hybrid or wgrad" > $OUT/${TAG}_pytest_hybrid.log 2>&1; echo "hybrid rc=$?"; tail -3 $OUT/${TAG}_pytest_hybrid.log
UVTG_TN_TOUCH=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "hybrid" > $OUT/${TAG}_pytest_hybrid_pf2.log 2>&1; echo "hybrid pf2 rc=$?"; tail -3 $OUT/${TAG}_pytest_hybrid_pf2.log
bash tools/ab_env.sh "UVTG_TN_TOUCH=2" "" 2>&1 | tee $OUT/${TAG}_ab_touch2.log
bash tools/ab_env.sh "UVTG_TN_TOUCH=3" "" 2>&1 | tee $OUT/${TAG}_ab_touch3.log
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|Error" $OUT/${TAG}_pytest_gpu.log | head -20
