#!/bin/bash
# A/B of two libuvtg.so builds on the same box: alternating bench.py runs (UVTG_LIB_PATH), prints the step and GEMM-family times
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OLD=${1:-$R/tools/libuvtg_old.so}
for round in 1 2; do
for lib in $OLD $R/univtg_amd/libuvtg.so; do
  UVTG_LIB_PATH=$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-padded-compare 2>/dev/null | tail -1 > /tmp/b.json
  python - "$lib" <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read())
g = d['roofline']['all_gemm_kernels']
print(f"{sys.argv[1].split('/')[-1]:18s} step {d['ms_per_step']:.3f} ms (event median {d['ms_per_step_event_median']:.3f}) enc {d['t_encoder_ms']:.3f} | nt256 {g['gemm_nt256_kernel']['ms_per_step']:.3f} ms {g['gemm_nt256_kernel']['tflops']:.0f} TF | tn {g['gemm_tn_kernel']['ms_per_step']:.3f} ms {g['gemm_tn_kernel']['tflops']:.0f} TF")
PY
done; done
