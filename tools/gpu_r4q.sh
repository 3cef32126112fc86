#!/bin/bash
# round 4, visit q: final build with the single-tile NT variants -- whole GPU suite, smoke, PMC passes (copied into profiles/ on the box so
# that the bench lines quote them), the bench lines, rocprofv3 kernel stats and step tables
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp; TAG=r04
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) > $OUT/smoke.log; cat $OUT/smoke.log
timeout 400 bash tools/pmc_nt256.sh > $OUT/${TAG}_pmc_nt256.log 2>&1; tail -22 $OUT/${TAG}_pmc_nt256.log | head -12
cp $OUT/pmc_nt256.json $OUT/${TAG}_pmc_nt256.json; cp $OUT/pmc_nt256.json $R/profiles/${TAG}_pmc_nt256.json; for i in 0 1 2; do cp $OUT/pmc_nt_$i.txt $OUT/${TAG}_pmc_nt_$i.txt; done
( timeout 400 python bench.py 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config2.json; cut -c1-200 $OUT/${TAG}_bench_config2.json
( timeout 200 python bench.py --variant B --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config2_variantB.json; cut -c1-160 $OUT/${TAG}_bench_config2_variantB.json
( timeout 400 python bench.py --mode infer 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_infer.json; cut -c1-200 $OUT/${TAG}_bench_infer.json
for c in 3 5 4; do ( timeout 150 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config$c.json; cut -c1-120 $OUT/${TAG}_bench_config$c.json; done
cd /tmp; rm -rf /tmp/st
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-companions > $OUT/${TAG}_stats_run.log 2>&1
cp $(find /tmp/st -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_rocprofv3_kernel_stats.csv; head -4 $OUT/${TAG}_rocprofv3_kernel_stats.csv | cut -c1-160
cd $R
timeout 300 bash tools/prof.sh ${TAG}c2 26 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-companions > /dev/null 2>&1
timeout 300 bash tools/prof.sh ${TAG}c2B 26 python $R/bench.py --variant B --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-companions > /dev/null 2>&1
timeout 300 bash tools/prof.sh ${TAG}c4 26 python $R/bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-companions > /dev/null 2>&1
timeout 300 bash tools/prof.sh ${TAG}inf32 40 python $R/tools/infer_prof.py 32 > /dev/null 2>&1
timeout 300 bash tools/prof.sh ${TAG}inf1 40 python $R/tools/infer_prof.py 1 > /dev/null 2>&1
ls $OUT | head -50
