#!/bin/bash
# generic round-5 visit: tools/gpu_visit.sh '<pytest -k expression or "">' then alternating A/B arms given as env AB_SPECS (newline-separated "<label>|<ENV=..>")
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
K="${1:-}"
if [ -n "$K" ]; then timeout 1800 python -m pytest tests -m gpu -q -x -k "$K" > $OUT/visit_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/visit_pytest.log | cut -c1-200; grep -n "FAILED\|^E  " $OUT/visit_pytest.log | head; fi
