#!/bin/bash
# round-6 visit a: full GPU suite (index-clause table recorded), smoke, default bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
rm -f $OUT/index_clause.txt
UVTG_INDEX_CLAUSE_RECORD=1 timeout 2400 python -m pytest tests -m gpu -q > $OUT/r06a_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r06a_pytest_gpu.log | cut -c1-300
grep -n "FAILED\|^E  " $OUT/r06a_pytest_gpu.log | head -20
cat $OUT/index_clause.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $OUT/r06a_smoke.log; cat $OUT/r06a_smoke.log
( timeout 1200 python bench.py 2>$OUT/r06a_bench.err | tail -1 ) > $OUT/r06a_bench_config2.json; cut -c1-300 $OUT/r06a_bench_config2.json
tail -5 $OUT/r06a_bench.err
