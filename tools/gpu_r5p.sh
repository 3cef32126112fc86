#!/bin/bash
# round 5, visit p: does delaying the P-wave's block (s_sleep) put its MFMAs beside the S-wave's VALU section?  attn_bench at S = 1232, alternating
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for round in 1 2; do
for p in 0 4 5 6 3 1; do
  echo -n "UVTG_ATTN_WS_PRIO=$p  "; UVTG_ATTN_WS_PRIO=$p python tools/attn_bench.py 2>/dev/null | head -1
done; done | tee $OUT/r5p_ws_skew.txt
