#!/bin/bash
# round 4, visit c: whole GPU suite (strict index rule with the margin criterion, hl production-width test), smoke, then the round's artifact set
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|^FAILED\|^\[config2 fp32\|^\[seeds\|   sample" $OUT/pytest_gpu.log | cut -c1-260 | head -40
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) > $OUT/smoke.log; cat $OUT/smoke.log
bash tools/artifacts_r04.sh r04
