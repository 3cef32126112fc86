#!/bin/bash
# visit T: fused attention backward variants (V fragments resident | requested at the top of the iteration) against the committed build
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
BASE=$R/univtg_amd/libuvtg_base.so
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "attention_bwd" 2>&1 | tail -2
line() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ra = d.get("roofline_attention", {})
print(f"   {d['ms_per_step']:.3f} ms/step  t_encoder {d.get('t_encoder_ms')}  attn fwd {ra.get('forward', {}).get('ms_per_step')} bwd {ra.get('backward', {}).get('ms_per_step')}")
PY
}
for cfg in 2 3; do
for arm in new base new base; do
  unset UVTG_LIB_PATH
  if [ $arm = base ]; then export UVTG_LIB_PATH=$BASE; fi
  timeout 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-padded-compare 2>/dev/null | tail -1 > /tmp/b.json
  echo "config $cfg $arm:"; line /tmp/b.json
done; done 2>&1 | tee $OUT/r03t_fused_v_resident.txt
