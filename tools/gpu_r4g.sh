#!/bin/bash
# round 4, visit g: PMC passes of the bf16 persistent NT GEMM on the final build (HALF instantiations excluded), then the headline and variant-B
# bench lines quoting them (the PMC summary is copied into profiles/ ON THE BOX first: bench.py checks its kernel-source hash)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 400 bash tools/pmc_nt256.sh > $OUT/r04_pmc_nt256.log 2>&1; tail -22 $OUT/r04_pmc_nt256.log | head -12
cp $OUT/pmc_nt256.json $OUT/r04_pmc_nt256.json; cp $OUT/pmc_nt256.json $R/profiles/r04_pmc_nt256.json; for i in 0 1 2; do cp $OUT/pmc_nt_$i.txt $OUT/r04_pmc_nt_$i.txt; done
( timeout 400 python bench.py 2>/dev/null | tail -1 ) > $OUT/r04_bench_config2.json; cut -c1-200 $OUT/r04_bench_config2.json
( timeout 200 python bench.py --variant B --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/r04_bench_config2_variantB.json; cut -c1-160 $OUT/r04_bench_config2_variantB.json
