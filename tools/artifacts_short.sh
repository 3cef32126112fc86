#!/bin/bash
# Round artifacts in priority order for a short GPU budget: config-2 bench line, rocprofv3 kernel stats of the bench command, PMC passes of
# the dominant GEMM, config 3 / 5 / 4 bench lines, per-step kernel table.  Everything lands in gpurun_out/; copy what is judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r02}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
( timeout 300 python bench.py 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config2.json
cut -c1-300 $OUT/${TAG}_bench_config2.json
cd /tmp; rm -rf /tmp/st
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-padded-compare > $OUT/${TAG}_stats_run.log 2>&1
cp $(find /tmp/st -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_rocprofv3_kernel_stats.csv
head -8 $OUT/${TAG}_rocprofv3_kernel_stats.csv | cut -c1-160
cd $R
timeout 240 bash tools/pmc_nt256.sh > $OUT/${TAG}_pmc_nt256.log 2>&1
cp $OUT/pmc_nt256.json $OUT/${TAG}_pmc_nt256.json; for i in 0 1 2; do cp $OUT/pmc_nt_$i.txt $OUT/${TAG}_pmc_nt_$i.txt; done
tail -22 $OUT/${TAG}_pmc_nt256.log | head -14
for c in 3 5 4; do ( timeout 120 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config$c.json; cut -c1-120 $OUT/${TAG}_bench_config$c.json; done
bash tools/prof.sh ${TAG}c2 26 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
