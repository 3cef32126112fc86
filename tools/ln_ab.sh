#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_model.py -m gpu -q -x > $OUT/pytest_ln.log 2>&1; tail -3 $OUT/pytest_ln.log
run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-padded-compare --profile-steps 0 ${CFG:-} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "generic $(UVTG_LN_LEAN_OFF=1 run)"
  echo "lean    $(run)"
done
bash tools/prof.sh lean 13 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
grep -n "ln_bwd\|ln_fwd" $OUT/lean_stats.md | cut -c1-150
