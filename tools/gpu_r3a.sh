#!/bin/bash
# round-3 visit A: whole GPU suite, smoke, bench lines (train B / train A / infer), FETCH_SIZE calibration + PMC passes, L2->LDS probe
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp; TAG=${1:-r03a}
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|Error\|^\[" $OUT/${TAG}_pytest_gpu.log | head -60
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $OUT/${TAG}_smoke.log; cat $OUT/${TAG}_smoke.log
( timeout 600 python bench.py 2>$OUT/${TAG}_bench.err | tail -1 ) > $OUT/${TAG}_bench_config2.json; cut -c1-400 $OUT/${TAG}_bench_config2.json; tail -3 $OUT/${TAG}_bench.err
( timeout 300 python bench.py --variant A --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config2_variantA.json; cut -c1-200 $OUT/${TAG}_bench_config2_variantA.json
( timeout 400 python bench.py --mode infer 2>$OUT/${TAG}_infer.err | tail -1 ) > $OUT/${TAG}_bench_infer.json; cut -c1-1500 $OUT/${TAG}_bench_infer.json; tail -3 $OUT/${TAG}_infer.err
timeout 200 bash tools/fetch_calib.sh > $OUT/${TAG}_fetch_calib.log 2>&1; tail -30 $OUT/${TAG}_fetch_calib.log
timeout 300 bash tools/pmc_nt256.sh > $OUT/${TAG}_pmc_nt256.log 2>&1; cp $OUT/pmc_nt256.json $OUT/${TAG}_pmc_nt256.json; tail -24 $OUT/${TAG}_pmc_nt256.log
timeout 120 $R/tools/l2_lds_bw > $OUT/${TAG}_l2_lds_bw.log 2>&1; cat $OUT/${TAG}_l2_lds_bw.log
