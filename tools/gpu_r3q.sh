#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/r03q_pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|^\[config\|^\[2-rank\|^\[conv" $OUT/r03q_pytest_gpu.log | head -30
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) | tee $OUT/r03q_smoke.log
bash tools/artifacts_r03.sh r03
