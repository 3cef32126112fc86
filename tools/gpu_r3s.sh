#!/bin/bash
# visit S: config 3 (S = 160): the 8-wave fused attention backward against the (now pipelined / LDS-DMA) split kernels
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
line() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ra = d.get("roofline_attention", {})
print(f"   {d['ms_per_step']:.3f} ms/step  t_encoder {d.get('t_encoder_ms')}  attn fwd {ra.get('forward', {}).get('ms_per_step')} bwd {ra.get('backward', {}).get('ms_per_step')}")
PY
}
for arm in fused8 split fused8 split; do
  unset UVTG_ATTN_FUSED8_OFF
  if [ $arm = split ]; then export UVTG_ATTN_FUSED8_OFF=1; fi
  timeout 300 python bench.py --config 3 --steps 30 --warmup 5 --no-cpu-baseline --no-padded-compare 2>/dev/null | tail -1 > /tmp/b.json
  echo "config 3 $arm:"; line /tmp/b.json
done 2>&1 | tee $OUT/r03s_config3_fused8_vs_split.txt
