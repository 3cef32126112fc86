#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 tools/gemm_pp_lab > $OUT/pp_lab.log 2>&1; echo "lab rc=$?"
cat $OUT/pp_lab.log
timeout 900 python -m pytest tests/test_gpu_parity_full.py -m gpu -q -s -rA -k "fp32x3 or known_answers or prefetcher" > $OUT/parity_full3.log 2>&1
grep -n "^\[\|passed\|failed\|^E " $OUT/parity_full3.log | head -40
