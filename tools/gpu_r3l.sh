#!/bin/bash
# visit L: attention backward rewrite (lean softmax sections, chunk-swizzled tiles, [key][query] dS image) -- LDS conflict probe, kernel tests,
# kernel timings new / padded rows / previous build, step A/B at configs 2 and 4
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
PREV=$R/univtg_amd/libuvtg_prev.so
timeout 60 tools/lds_conflict_probe 2>&1 | tee $OUT/r03l_lds_conflict_probe.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "attention" 2>&1 | tail -5
echo "== attn_bench: new"; timeout 200 python tools/attn_bench.py 2>&1 | tee $OUT/r03l_attn_new.txt
echo "== attn_bench: new, padded rows"; UVTG_ATTN_SWZ_OFF=1 timeout 200 python tools/attn_bench.py 2>&1 | tee $OUT/r03l_attn_new_padded.txt
echo "== attn_bench: previous build"; UVTG_LIB_PATH=$PREV timeout 200 python tools/attn_bench.py 2>&1 | tee $OUT/r03l_attn_prev.txt
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -5
line() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ra = d.get("roofline_attention", {})
print(f"   {d['ms_per_step']:.3f} ms/step  t_encoder {d.get('t_encoder_ms')}  attn fwd {ra.get('forward', {}).get('ms_per_step')} bwd {ra.get('backward', {}).get('ms_per_step')}")
PY
}
for cfg in 2 4; do
  for arm in new prev new prev; do
    if [ $arm = prev ]; then export UVTG_LIB_PATH=$PREV; else unset UVTG_LIB_PATH; fi
    timeout 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-padded-compare 2>/dev/null | tail -1 > /tmp/b.json
    echo "config $cfg $arm:"; line /tmp/b.json
  done
done 2>&1 | tee $OUT/r03l_step_ab.txt
unset UVTG_LIB_PATH
