#!/bin/bash
# round-6 visit c: grouped deferral under readiness events (tests in both modes), the side-by-side epilogue-overlap probe, A/B of the N > 1 step on one rank
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -k "overlapped or two_rank or native_train_step or conv_head_weight or loader_waves or dropout_replayed_through_oracle" > $OUT/r06c_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r06c_pytest.log | cut -c1-300
grep -n "FAILED\|^E  " $OUT/r06c_pytest.log | head -20
UVTG_TN_EVENTS_PER_LAYER=1 timeout 900 python -m pytest tests -m gpu -q -x -k "overlapped or two_rank" > $OUT/r06c_pytest_per_layer.log 2>&1; echo "pytest (per-layer events) rc=$?"; tail -2 $OUT/r06c_pytest_per_layer.log | cut -c1-300
timeout 600 python tools/epilogue_overlap_probe.py > $OUT/r06c_epilogue_overlap_probe.txt 2>&1; tail -5 $OUT/r06c_epilogue_overlap_probe.txt
bash tools/ab5.sh 2 "single-rank step|" "N>1 step on one rank, grouped deferral||--overlap force" "N>1 step, per-layer events (r5)|UVTG_TN_EVENTS_PER_LAYER=1|--overlap force" "N>1 step, one launch behind the loop|UVTG_TN_DEFER_EVENTS=1|--overlap force" > $OUT/r06c_ab_multi_rank_step.txt 2>&1; cat $OUT/r06c_ab_multi_rank_step.txt
