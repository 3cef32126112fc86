#!/bin/bash
# visit W: what reserving CUs for the communication kernels costs a single-GPU step (UVTG_COMM_CUS = 0 / 8 / 16 / 32)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for k in 0 8 16 32 0; do
  timeout 300 python tools/reserved_cus_probe.py $k --steps 30 --warmup 5 --no-cpu-baseline --no-padded-compare 2>/dev/null | tail -1 > /tmp/b.json
  python - $k <<'PY'
import json, sys
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print(f"UVTG_COMM_CUS={sys.argv[1]:>2s}: {d['ms_per_step']:.3f} ms/step  t_encoder {d.get('t_encoder_ms')}  nt256 {d['roofline']['all_gemm_kernels']['gemm_nt256_kernel']['tflops']} TF  tn {d['roofline']['all_gemm_kernels']['gemm_tn_kernel']['tflops']} TF")
PY
done 2>&1 | tee $OUT/r03w_reserved_cus.txt
