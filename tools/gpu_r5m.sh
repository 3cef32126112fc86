#!/bin/bash
# round 5, visit m: what the role-split dK / dV kernel waits for -- per-kernel times of tools/attn_bench.py (full S = 1232, no raggedness) and one
# rocprofv3 --pmc pass (SQ counters) over it, role-split vs one-wave-per-SIMD
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
python tools/attn_bench.py 2>/dev/null | tee $OUT/r5m_attn_bench_ws.txt
UVTG_ATTN_WS_OFF=1 python tools/attn_bench.py 2>/dev/null | tee $OUT/r5m_attn_bench_old.txt
export PMC_EXTRA="--kernel-include-regex attn_bwd_dkdv"
bash tools/pmc.sh attnws "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" -- python $R/tools/attn_bench.py > /dev/null
cp $OUT/pmc_attnws_0.txt $OUT/r5m_pmc_dkdv_ws_0.txt; cp $OUT/pmc_attnws_1.txt $OUT/r5m_pmc_dkdv_ws_1.txt
UVTG_ATTN_WS_OFF=1 bash tools/pmc.sh attnold "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- python $R/tools/attn_bench.py > /dev/null
cp $OUT/pmc_attnold_0.txt $OUT/r5m_pmc_dkdv_old_0.txt
cd /tmp; rm -rf /tmp/st2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st2 -o ab -- python $R/tools/attn_bench.py > /dev/null 2>&1
cut -d, -f1-7 $(find /tmp/st2 -name '*kernel_stats.csv' | head -1) | head -12 | tee $OUT/r5m_attn_bench_kernel_stats.csv
