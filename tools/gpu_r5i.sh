#!/bin/bash
# round 5, visit i: product waves own a head-dim block (8 transposing reads per wave and block instead of 32) -- tests + A/B against the key-group
# version and the one-wave-per-SIMD kernel, configs 4 / 5, same box
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "role_split or attention_bwd" > $OUT/r5i_pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/r5i_pytest.log | cut -c1-200
UVTG_ATTN_WS_KEYP=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "role_split" > $OUT/r5i_pytest_keyp.log 2>&1; echo "pytest keyp rc=$?"; tail -1 $OUT/r5i_pytest_keyp.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q -x -k "config4 or config5 or dropout" > $OUT/r5i_pytest_c4.log 2>&1; echo "pytest config4 rc=$?"; tail -1 $OUT/r5i_pytest_c4.log | cut -c1-200
AB_ARGS="--config 4" bash tools/ab5.sh 2 "c4 one-wave-per-SIMD dK/dV|UVTG_ATTN_WS_OFF=1" "c4 role split, P-waves by key group|UVTG_ATTN_WS_KEYP=1" "c4 role split, P-waves by head-dim block (default)|" 2>&1 | tee $OUT/r5i_ab.txt
AB_ARGS="--config 5" bash tools/ab5.sh 1 "c5 one-wave-per-SIMD dK/dV|UVTG_ATTN_WS_OFF=1" "c5 role split (default)|" 2>&1 | tee -a $OUT/r5i_ab.txt
