#!/bin/bash
# visit R: stability of the GPU suite (two more full runs on a fresh box; the margins of the tightest floors are printed)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for i in 1 2; do
  timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/r03r_pytest_gpu_$i.log 2>&1; echo "run $i pytest rc=$?"
  grep -n "passed\|failed\|^E  \|worst cosine\|identical ranking\|top-1\|per-step relative gap" $OUT/r03r_pytest_gpu_$i.log | cut -c1-330 | head -20
done
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 )
timeout 300 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 | cut -c1-400
