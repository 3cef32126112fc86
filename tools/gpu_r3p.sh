#!/bin/bash
# visit P: ablation of the long-sequence dK/dV kernel (measurement build)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
export UVTG_LIB_PATH=$R/univtg_amd/libuvtg_abl.so
for m in 6 1 2 3 4 5 6; do UVTG_ATTN_ABL=$m timeout 120 python tools/attn_abl.py 2>&1 | grep "UVTG_ATTN_ABL"; done | tee $OUT/r03p_dkdv_ablation.txt
