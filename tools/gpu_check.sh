#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench line, rocprofv3 kernel stats.  Outputs under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > $OUT/smoke.log
( timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -3 ) > $OUT/bench.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r1 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > $OUT/prof_run.log 2>&1
cd $R
find $OUT/prof -name '*kernel_stats*' | head
cat $OUT/pytest_gpu.log $OUT/smoke.log $OUT/bench.log
