#!/bin/bash
# Short GPU visit: whole GPU suite, smoke, config-2 bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|Error" $OUT/pytest_gpu.log | head -40
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $OUT/smoke.log; cat $OUT/smoke.log
( timeout 600 python bench.py 2>/dev/null | tail -1 ) > $OUT/bench_c2.json; cut -c1-300 $OUT/bench_c2.json
