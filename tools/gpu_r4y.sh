#!/bin/bash
# round 4, visit y (last): the two bench lines again with the final bench.py (clock note beside the roofline); same kernel source as the PMC summary in profiles/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp; TAG=r04
( timeout 400 python bench.py 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config2.json; cut -c1-200 $OUT/${TAG}_bench_config2.json
( timeout 200 python bench.py --variant B --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/${TAG}_bench_config2_variantB.json; cut -c1-160 $OUT/${TAG}_bench_config2_variantB.json
python -c "
import json; d=json.loads(open('$OUT/${TAG}_bench_config2.json').read()); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['traffic'], r['traffic_note'][:60], '|', r['clock_note'][:50])"
