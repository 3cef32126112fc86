#!/bin/bash
# A/B of two environment settings on the same box: alternating bench.py runs, prints step / encoder / GEMM-family times
#   tools/ab_env.sh "UVTG_TN_DEFER_OFF=1" "" [extra bench.py args...]      ("" = defaults)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
A="$1"; B="$2"; shift 2
for round in 1 2; do
for E in "$A" "$B"; do
  env UVTG_DEV_ENV=1 $E timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-padded-compare "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python - "${E:-default}" <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read())
g = d['roofline']['all_gemm_kernels']
print(f"{sys.argv[1]:28s} step {d['ms_per_step']:.3f} ms (event median {d['ms_per_step_event_median']:.3f}) enc {d['t_encoder_ms']:.3f} fwd/bwd {d['sections']['forward_ms']:.3f}/{d['sections']['backward_ms']:.3f} | nt256 {g['gemm_nt256_kernel']['ms_per_step']:.3f} ms {g['gemm_nt256_kernel']['tflops']:.0f} TF ({g['gemm_nt256_kernel']['launches_per_step']}) | tn {g['gemm_tn_kernel']['ms_per_step']:.3f} ms {g['gemm_tn_kernel']['tflops']:.0f} TF ({g['gemm_tn_kernel']['launches_per_step']})")
PY
done; done
