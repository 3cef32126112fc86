#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for E in "" "UVTG_TN_EVENT_GROUPS=1" "UVTG_TN_EVENTS_PER_LAYER=1"; do
  env UVTG_DEV_ENV=1 $E timeout 1500 python -m pytest tests -m gpu -q -x -k "overlapped or two_rank or ddp_wrapper" > $OUT/r06h_pytest.log 2>&1; echo "pytest [$E] rc=$?"; grep -n "passed\|failed" $OUT/r06h_pytest.log | tail -1; grep -n "^FAILED\|^E  " $OUT/r06h_pytest.log | head -5
done
bash tools/ab5.sh 2 "single-rank step|" "N>1 on one rank (default: one deferred launch, coalesced exchange)||--overlap force" "N>1, two readiness groups|UVTG_TN_EVENT_GROUPS=1|--overlap force" "N>1, per-layer events + one collective per range (r5)|UVTG_TN_EVENTS_PER_LAYER=1 UVTG_COMM_COALESCE_OFF=1|--overlap force" > $OUT/r06h_ab_coalesced.txt 2>&1; cat $OUT/r06h_ab_coalesced.txt
