#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/r03e_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|^E  " $OUT/r03e_pytest_gpu.log | head
bash tools/ab_env.sh "UVTG_NT_OLD_128_CHOICE=1" "" 2>&1 | tee $OUT/r03e_ab_128choice.log
bash tools/artifacts_r03.sh r03
