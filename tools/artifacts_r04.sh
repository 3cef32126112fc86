#!/bin/bash
# Round-4 artifact set (everything lands in gpurun_out/; copy what is judged into profiles/): headline bench line (variant A, precise projections,
# CPU baseline), variant-B line, rocprofv3 kernel stats of the bench command, PMC passes of the dominant GEMM, inference line, configs 3 / 5 / 4,
# per-step kernel table of config 2
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r04}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
( timeout 400 python bench.py 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config2.json; cut -c1-200 $OUT/${TAG}_bench_config2.json
( timeout 200 python bench.py --variant B --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config2_variantB.json; cut -c1-160 $OUT/${TAG}_bench_config2_variantB.json
cd /tmp; rm -rf /tmp/st
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-companions > $OUT/${TAG}_stats_run.log 2>&1
cp $(find /tmp/st -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_rocprofv3_kernel_stats.csv; head -6 $OUT/${TAG}_rocprofv3_kernel_stats.csv | cut -c1-160
cd $R
timeout 300 bash tools/pmc_nt256.sh > $OUT/${TAG}_pmc_nt256.log 2>&1
cp $OUT/pmc_nt256.json $OUT/${TAG}_pmc_nt256.json; for i in 0 1 2; do cp $OUT/pmc_nt_$i.txt $OUT/${TAG}_pmc_nt_$i.txt; done
tail -24 $OUT/${TAG}_pmc_nt256.log | head -14
( timeout 400 python bench.py --mode infer 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_infer.json; cut -c1-200 $OUT/${TAG}_bench_infer.json
for c in 3 5 4; do ( timeout 150 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config$c.json; cut -c1-120 $OUT/${TAG}_bench_config$c.json; done
timeout 400 bash tools/prof.sh ${TAG}c2 26 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-companions > /dev/null 2>&1
