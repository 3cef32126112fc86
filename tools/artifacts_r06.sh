#!/bin/bash
# Round-6 artifact set (round 5: the same script, tag r05) (lands in gpurun_out/; what is judged is copied into profiles/): GPU suite + smoke, PMC passes of the dominant GEMM (copied
# into profiles/ ON THE BOX so that the bench line that follows quotes them), the default bench line (reference CPU baseline, every config as a
# companion), rocprofv3 --kernel-trace --stats of the bench command, per-step kernel tables of configs 2 and 4, the inference line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/${TAG}_pytest_gpu.log | cut -c1-200
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $OUT/${TAG}_smoke.log; cat $OUT/${TAG}_smoke.log
timeout 400 bash tools/pmc_nt256.sh > $OUT/${TAG}_pmc_nt256.log 2>&1
cp $OUT/pmc_nt256.json $OUT/${TAG}_pmc_nt256.json; cp $OUT/pmc_nt256.json $R/profiles/${TAG}_pmc_nt256.json
for i in 0 1 2; do cp $OUT/pmc_nt_$i.txt $OUT/${TAG}_pmc_nt_$i.txt; done
tail -24 $OUT/${TAG}_pmc_nt256.log | head -16
( timeout 900 python bench.py 2>$OUT/${TAG}_bench.err | tail -1 ) > $OUT/${TAG}_bench_config2.json; cut -c1-200 $OUT/${TAG}_bench_config2.json
cd /tmp; rm -rf /tmp/st
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-companions > $OUT/${TAG}_stats_run.log 2>&1
cp $(find /tmp/st -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_rocprofv3_kernel_stats.csv; head -6 $OUT/${TAG}_rocprofv3_kernel_stats.csv | cut -c1-160
cd $R
timeout 400 bash tools/prof.sh ${TAG}c2 26 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-companions > /dev/null 2>&1
timeout 400 bash tools/prof.sh ${TAG}c4 26 python $R/bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-companions > /dev/null 2>&1
( timeout 400 python bench.py --mode infer 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_infer.json; cut -c1-200 $OUT/${TAG}_bench_infer.json
