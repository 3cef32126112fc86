#!/bin/bash
# round 5, visit g: role-split dK / dV kernel (head_dim 128, long sequences) -- attention tests, config-4 model tests, config-4 A/B, kernel table
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" > $OUT/r5g_pytest_attn.log 2>&1; echo "pytest attn rc=$?"
tail -3 $OUT/r5g_pytest_attn.log | cut -c1-300; grep -n "^E  " $OUT/r5g_pytest_attn.log | head
timeout 1500 python -m pytest tests -m gpu -q -x -k "config4 or config5 or dropout" > $OUT/r5g_pytest_c4.log 2>&1; echo "pytest config4 rc=$?"
tail -3 $OUT/r5g_pytest_c4.log | cut -c1-300; grep -n "^E  " $OUT/r5g_pytest_c4.log | head
AB_ARGS="--config 4" bash tools/ab5.sh 2 "config 4 default (role-split dK/dV)|" "config 4 one-wave-per-SIMD dK/dV|UVTG_ATTN_WS_OFF=1" 2>&1 | tee $OUT/r5g_ab.txt
AB_ARGS="--config 5" bash tools/ab5.sh 1 "config 5 default (role-split dK/dV)|" "config 5 one-wave-per-SIMD dK/dV|UVTG_ATTN_WS_OFF=1" 2>&1 | tee -a $OUT/r5g_ab.txt
timeout 400 bash tools/prof.sh r5gc4 23 python $R/bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-companions > /dev/null 2>&1
head -30 $OUT/r5gc4_stats.md | cut -c1-200
