#!/bin/bash
# The round as ONE in-box A/B: every round-5 path switched back by its experiment switch against the defaults, alternating on one box
# (profiles/r05_ab_whole_round.txt).  Run through gpurun: bash tools/ab_whole_round.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
OLD="UVTG_LN_FWD_LEAN_OFF=1 UVTG_LN_DEFER_OFF=1 UVTG_DELTA_FUSE_OFF=1 UVTG_NT_PLAN_OVR=27392,3072,192,0,0 UVTG_ATTN_WS_OFF=1 UVTG_ATTN_SAMPLE_MAJOR=1 UVTG_ATTN_FWD_DMA_OFF=1 UVTG_LAST_CLIP_OFF=1 UVTG_TN_CONV_DEFER_OFF=1"
{
bash tools/ab5.sh 2 "config 2 (headline): round-4 paths|$OLD" "config 2 (headline): round-5 defaults|"
AB_ARGS="--variant B" bash tools/ab5.sh 1 "config 2 variant B: round-4 paths|$OLD" "config 2 variant B: round-5 defaults|"
for c in 3 4 5; do AB_ARGS="--config $c" bash tools/ab5.sh 1 "config $c: round-4 paths|$OLD" "config $c: round-5 defaults|"; done
} 2>&1 | tee gpurun_out/ab_whole_round.txt
