#!/bin/bash
# round 5, visit k: role-split dK / dV with the Q / dO rows requested two iterations ahead -- tests + A/B (config 4, same box)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "role_split or attention_bwd" > $OUT/r5k_pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/r5k_pytest.log | cut -c1-200
UVTG_ATTN_WS_PF1=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "role_split" > $OUT/r5k_pytest_pf1.log 2>&1; echo "pytest pf1 rc=$?"; tail -1 $OUT/r5k_pytest_pf1.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q -x -k "config4 or config5 or dropout" > $OUT/r5k_pytest_c4.log 2>&1; echo "pytest config4 rc=$?"; tail -1 $OUT/r5k_pytest_c4.log | cut -c1-200
AB_ARGS="--config 4" bash tools/ab5.sh 2 "c4 one-wave-per-SIMD dK/dV|UVTG_ATTN_WS_OFF=1" "c4 role split, rows one iteration ahead|UVTG_ATTN_WS_PF1=1" "c4 role split, rows two iterations ahead (default)|" 2>&1 | tee $OUT/r5k_ab.txt
