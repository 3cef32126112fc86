#!/bin/bash
# round 5, visit h: work split of the role-split dK / dV kernel -- modes 0 / 1 / 2 against the one-wave-per-SIMD kernel, config 4 (and 5), same box
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for m in 0 1 2; do
  UVTG_ATTN_WS_MODE=$m timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "role_split or attention_bwd" > $OUT/r5h_pytest_m$m.log 2>&1; echo "mode $m pytest rc=$?"
  tail -1 $OUT/r5h_pytest_m$m.log | cut -c1-200
done
AB_ARGS="--config 4" bash tools/ab5.sh 2 "c4 one-wave-per-SIMD dK/dV|UVTG_ATTN_WS_OFF=1" "c4 role split, mode 0 (S: all softmax)|UVTG_ATTN_WS_MODE=0" "c4 role split, mode 1 (P: dP, dS)|UVTG_ATTN_WS_MODE=1" "c4 role split, mode 2 (S: dP; P: dS)|UVTG_ATTN_WS_MODE=2" 2>&1 | tee $OUT/r5h_ab.txt
AB_ARGS="--config 5" bash tools/ab5.sh 1 "c5 one-wave-per-SIMD dK/dV|UVTG_ATTN_WS_OFF=1" "c5 role split, mode 0|UVTG_ATTN_WS_MODE=0" "c5 role split, mode 2|UVTG_ATTN_WS_MODE=2" 2>&1 | tee -a $OUT/r5h_ab.txt
