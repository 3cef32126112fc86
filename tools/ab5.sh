#!/bin/bash
# in-box A/B on alternating bench runs (headline workload): tools/ab5.sh <rounds> "<label>|<ENV=val ENV2=val ...>[|<extra bench.py arguments>]" ...   (empty env = defaults)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
ROUNDS=$1; shift
for round in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    L="${spec%%|*}"; E="${spec#*|}"; X=""
    case "$E" in *"|"*) X="${E#*|}"; E="${E%%|*}";; esac
    env UVTG_DEV_ENV=1 $E timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-companions $AB_ARGS $X 2>/dev/null | grep "^{\"metric\"" | tail -1 > /tmp/b.json
    python - "$L" <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read())
g = d['roofline']['all_gemm_kernels']; h = d['roofline_hbm']; a = d['roofline_attention']
print(f"{sys.argv[1]:40s} step {d['ms_per_step']:.3f} (med {d['ms_per_step_event_median']:.3f}) enc {d['t_encoder_ms']:.3f} f/b {d['sections']['forward_ms']:.3f}/{d['sections']['backward_ms']:.3f} | nt256 {g['gemm_nt256_kernel']['ms_per_step']:.3f} {g['gemm_nt256_kernel']['tflops']:.0f}TF ({g['gemm_nt256_kernel']['launches_per_step']}) tn {g['gemm_tn_kernel']['ms_per_step']:.3f} | ln f/b {h['layernorm_forward']['ms_per_step']:.3f}/{h['layernorm_backward']['ms_per_step']:.3f} attn f/b {a['forward']['ms_per_step']:.3f}/{a['backward']['ms_per_step']:.3f}")
PY
  done
done
