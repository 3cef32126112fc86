#!/bin/bash
# round-6 visit d: full GPU suite on the dev-config build, LayerNorm backward row slots A/B (previous norm.hip as a second .so), default bench line (input-pipeline companion)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
rm -f $OUT/index_clause.txt
timeout 2400 python -m pytest tests -m gpu -q > $OUT/r06d_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/r06d_pytest_gpu.log | cut -c1-300
grep -n "^FAILED\|^E  " $OUT/r06d_pytest_gpu.log | head -20
bash tools/ab5.sh 2 "ln_bwd 2 row slots (new default)|" "ln_bwd previous kernel (1 prefetched row)|UVTG_LIB_PATH=$R/univtg_amd/libuvtg_prev.so" "ln_bwd 3 row slots (spills)|UVTG_LN_BWD_SLOTS=3" > $OUT/r06d_ab_ln_bwd_slots.txt 2>&1; cat $OUT/r06d_ab_ln_bwd_slots.txt
( timeout 1200 python bench.py --no-other-configs 2>$OUT/r06d_bench.err | tail -1 ) > $OUT/r06d_bench_config2.json; cut -c1-200 $OUT/r06d_bench_config2.json; tail -3 $OUT/r06d_bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06d_bench_config2.json'))
print(json.dumps(d['companions']['with_input_pipeline'], indent=1)[:2500])
print(json.dumps(d['companions']['drop_in_autograd'], indent=1)[:600])
PY
