#!/bin/bash
# staging-order experiment: whole-step bench per UVTG_NT_ORD setting (TM = 2, 3, 4), two rounds, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2; do
for ord in 111 101 000 011; do
  UVTG_NT_ORD=$ord timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-padded-compare 2>/dev/null | tail -1 > /tmp/b.json
  python - "$ord" <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read())
g = d['roofline']['all_gemm_kernels']
print(f"ORD {sys.argv[1]} step {d['ms_per_step']:.3f} ms enc {d['t_encoder_ms']:.3f} | nt256 {g['gemm_nt256_kernel']['ms_per_step']:.3f} ms {g['gemm_nt256_kernel']['tflops']:.0f} TF | tn {g['gemm_tn_kernel']['ms_per_step']:.3f} ms {g['gemm_tn_kernel']['tflops']:.0f} TF")
PY
done; done
