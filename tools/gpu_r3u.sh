#!/bin/bash
# visit U: ablation of the fused (S <= 128) attention backward (measurement build)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
export UVTG_LIB_PATH=$R/univtg_amd/libuvtg_abl.so
for m in 0 1 2 3 4 5 6 7 0; do UVTG_ATTN_FABL=$m timeout 120 python tools/attn_fabl.py 2>&1 | grep "UVTG_ATTN_FABL"; done | tee $OUT/r03u_fused_ablation.txt
