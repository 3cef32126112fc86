#!/bin/bash
# round 6, visit i: GPU suite + per-kernel table of the step (head pass with grouped loads) + step A/B against the previous library (tools/libuvtg_prev.so)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
UVTG_INDEX_CLAUSE_RECORD=1 timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/visit_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/visit_pytest.log | tail -2 | cut -c1-200; grep -n "FAILED\|^E  " $OUT/visit_pytest.log | head
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-companions --no-other-configs"
UVTG_DEV_ENV=1 bash tools/prof.sh step_new 20 $BENCH > /dev/null 2>&1
grep "heads_saliency_fwd\|wide_wave\|saliency_rows\|heads_final_bwd\|saliency_dq\|loss_\|seq_prep" $OUT/step_new_stats.md | cut -d'|' -f2-7 | cut -c1-200
if [ -f tools/libuvtg_prev.so ]; then bash tools/ab5.sh 4 "new|" "prev|UVTG_LIB_PATH=$R/tools/libuvtg_prev.so" 2>&1 | tee $OUT/ab_tail_kernels.txt; fi
