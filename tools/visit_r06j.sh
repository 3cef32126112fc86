#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/visit_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/visit_pytest.log | tail -2 | cut -c1-200; grep -n "FAILED\|^E  " $OUT/visit_pytest.log | head
AB_ARGS="--config 4" bash tools/ab5.sh 3 "rows split over 8 blocks per sample|" "one block per sample|UVTG_HEADS_DW_SPLIT=1" 2>&1 | tee $OUT/ab_heads_dw_split.txt
UVTG_DEV_ENV=1 bash tools/prof.sh c4new 20 python $R/bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-companions --no-other-configs > /dev/null 2>&1
grep "heads_final_bwd_dw\|heads_final_dw_reduce" $OUT/c4new_stats.md | cut -d'|' -f2-7 | cut -c1-200
