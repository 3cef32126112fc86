#!/bin/bash
# round 5, visit r: the round's kernel changes as ONE in-box A/B -- every round-5 path switched back to its round-4 behaviour by its experiment
# switch against the defaults, alternating, configs 2 (headline, variant A), 4, 5 and 2 variant B.  (Boxes differ by up to 8 % in wall clock:
# 8.88 - 9.58 ms for the same build on the visits of this round; only in-box ratios mean anything.)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "loader_waves" > $OUT/r5r_pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/r5r_pytest.log | cut -c1-200
OLD="UVTG_LN_FWD_LEAN_OFF=1 UVTG_LN_DEFER_OFF=1 UVTG_DELTA_FUSE_OFF=1 UVTG_NT_PLAN_OVR=27392,3072,192,0,0 UVTG_ATTN_WS_OFF=1 UVTG_ATTN_SAMPLE_MAJOR=1"
bash tools/ab5.sh 3 "config 2 (headline): round-4 paths|$OLD" "config 2 (headline): round-5 defaults|" 2>&1 | tee $OUT/r5r_ab_round.txt
AB_ARGS="--variant B" bash tools/ab5.sh 2 "config 2 variant B: round-4 paths|$OLD" "config 2 variant B: round-5 defaults|" 2>&1 | tee -a $OUT/r5r_ab_round.txt
AB_ARGS="--config 4" bash tools/ab5.sh 2 "config 4: round-4 paths|$OLD" "config 4: round-5 defaults|" 2>&1 | tee -a $OUT/r5r_ab_round.txt
AB_ARGS="--config 5" bash tools/ab5.sh 2 "config 5: round-4 paths|$OLD" "config 5: round-5 defaults|" 2>&1 | tee -a $OUT/r5r_ab_round.txt
AB_ARGS="--config 3" bash tools/ab5.sh 2 "config 3: round-4 paths|$OLD" "config 3: round-5 defaults|" 2>&1 | tee -a $OUT/r5r_ab_round.txt
