#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -k "overlapped or two_rank or ddp_wrapper" > $OUT/r06h_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/r06h_pytest.log | tail -1; grep -n "^FAILED\|^E  " $OUT/r06h_pytest.log | head
bash tools/ab5.sh 2 "single-rank step|" "N>1 on one rank, coalesced collectives per readiness group||--overlap force" "N>1 on one rank, one collective per range (r5)|UVTG_COMM_COALESCE_OFF=1|--overlap force" "N>1, one launch behind the loop, coalesced|UVTG_TN_DEFER_EVENTS=1|--overlap force" > $OUT/r06h_ab_coalesced.txt 2>&1; cat $OUT/r06h_ab_coalesced.txt
