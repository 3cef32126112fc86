#!/bin/bash
# visit O: where do the tiled attention kernels' operand bytes come from?  FETCH_SIZE (beyond-L2 traffic) + L2 hit counters per kernel, new build vs previous
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
PREV=$R/univtg_amd/libuvtg_prev.so
export PMC_EXTRA="--kernel-include-regex attn"
export PMC_TIMEOUT=120
bash tools/pmc.sh attn_new "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" -- python $R/tools/attn_bench.py > /dev/null
UVTG_LIB_PATH=$PREV bash tools/pmc.sh attn_prev "FETCH_SIZE" -- python $R/tools/attn_bench.py > /dev/null
tail -2 $OUT/pmc_attn_new_1.log | cut -c1-200
