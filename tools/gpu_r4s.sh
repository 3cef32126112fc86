#!/bin/bash
# round 4, visit s: where does the K split of the single-tile variant pay?  batch sweep x tile-count threshold (in-box, alternating)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
export INFER_AB_BATCHES=1,2,4,8,16,32 INFER_AB_PRECS=auto
{
for i in 1 2; do
for t in 0 8 16 32 64 80 128; do
  echo "split only launches of <= $t tiles: $(UVTG_NT_SPLITK_MAX_TILES=$t timeout 300 python tools/infer_ab.py 2>&1 | tail -1)"
done
done
} | tee $OUT/r04_ab_nt_split_k_threshold.txt
