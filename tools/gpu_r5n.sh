#!/bin/bash
# round 5, visit n: role-split dK / dV after the vmcnt fix -- tests, attn_bench, PMC, config 4 / 5 A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" > $OUT/r5n_pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/r5n_pytest.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q -x -k "config4 or config5 or dropout" > $OUT/r5n_pytest_c4.log 2>&1; echo "pytest configs rc=$?"; tail -1 $OUT/r5n_pytest_c4.log | cut -c1-200
python tools/attn_bench.py 2>/dev/null | tee $OUT/r5n_attn_bench_ws.txt
UVTG_ATTN_WS_PF1=1 python tools/attn_bench.py 2>/dev/null | head -1 | tee $OUT/r5n_attn_bench_ws_pf1.txt
UVTG_ATTN_WS_OFF=1 python tools/attn_bench.py 2>/dev/null | head -1 | tee $OUT/r5n_attn_bench_old.txt
export PMC_EXTRA="--kernel-include-regex attn_bwd_dkdv"
bash tools/pmc.sh attnws "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE" -- python $R/tools/attn_bench.py > /dev/null
cat $OUT/pmc_attnws_0.txt $OUT/pmc_attnws_1.txt > $OUT/r5n_pmc_dkdv_ws.txt
unset PMC_EXTRA
AB_ARGS="--config 4" bash tools/ab5.sh 2 "c4 one-wave-per-SIMD dK/dV|UVTG_ATTN_WS_OFF=1" "c4 role split (default)|" 2>&1 | tee $OUT/r5n_ab.txt
AB_ARGS="--config 5" bash tools/ab5.sh 1 "c5 one-wave-per-SIMD dK/dV|UVTG_ATTN_WS_OFF=1" "c5 role split (default)|" 2>&1 | tee -a $OUT/r5n_ab.txt
