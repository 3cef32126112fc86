#!/bin/bash
# round 4, visit v: PMC passes with the kernel filter pinned to the HALF argument, then the two bench lines that quote them (same build as visit q)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp; TAG=r04
timeout 400 bash tools/pmc_nt256.sh > $OUT/${TAG}_pmc_nt256.log 2>&1; tail -22 $OUT/${TAG}_pmc_nt256.log | head -12
cp $OUT/pmc_nt256.json $OUT/${TAG}_pmc_nt256.json; cp $OUT/pmc_nt256.json $R/profiles/${TAG}_pmc_nt256.json; for i in 0 1 2; do cp $OUT/pmc_nt_$i.txt $OUT/${TAG}_pmc_nt_$i.txt; done
( timeout 400 python bench.py 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config2.json; cut -c1-200 $OUT/${TAG}_bench_config2.json
( timeout 200 python bench.py --variant B --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config2_variantB.json; cut -c1-160 $OUT/${TAG}_bench_config2_variantB.json
head -30 $OUT/pmc_nt_0.txt | cut -c1-100
