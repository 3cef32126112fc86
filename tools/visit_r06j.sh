#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/visit_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/visit_pytest.log | tail -2 | cut -c1-200; grep -n "FAILED\|^E  " $OUT/visit_pytest.log | head
for c in 3 4 5; do
UVTG_DEV_ENV=1 bash tools/prof.sh c${c}new 20 python $R/bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-companions --no-other-configs > /dev/null 2>&1
echo "== config $c"; sed -n 5p $OUT/c${c}new_stats.md; grep -v "gemm_nt256\|gemm_tn256h" $OUT/c${c}new_stats.md | sed -n 9,40p | cut -d'|' -f2-7 | cut -c1-150
done
