set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for i in 1 2; do timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/r06_flake_$i.log 2>&1; echo "run $i rc=$?"; grep -n "passed\|failed" $OUT/r06_flake_$i.log | tail -1; done
SECONDS=0; timeout 1200 python bench.py > $OUT/r06_bench_timing.json 2>/dev/null; echo "default bench.py wall clock: $SECONDS s"; cut -c1-150 $OUT/r06_bench_timing.json
