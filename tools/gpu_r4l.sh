#!/bin/bash
# round 4, visit l: NT tile-height cost factors re-fitted on the round-4 trace (F3 1.07 -> 1.15, F5 1.02 -> 1.00): in-box A/B of the launch plans
# (univtg_amd/libuvtg_prev.so here = the ALTERNATIVE factors, libuvtg.so = the committed ones), variant A, variant B, config 3
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
ab() {
  for round in 1 2; do
  for lib in $R/univtg_amd/libuvtg.so $R/univtg_amd/libuvtg_prev.so; do
    UVTG_LIB_PATH=$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-companions "$@" 2>/dev/null | tail -1 > /tmp/b.json
    python - "$lib" "$*" <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read())
g = d['roofline']['all_gemm_kernels']
tag = "alt factors" if "prev" in sys.argv[1] else "committed  "
print(f"{sys.argv[2]:12s} {tag} step {d['ms_per_step']:.3f} ms (median {d['ms_per_step_event_median']:.3f}) enc {d['t_encoder_ms']:.3f} | nt256 {g['gemm_nt256_kernel']['ms_per_step']:.3f} ms {g['gemm_nt256_kernel']['tflops']:.0f} TF ({g['gemm_nt256_kernel']['launches_per_step']})")
PY
  done; done
}
( ab; ab --variant B; ab --config 3 ) | tee $OUT/r04_ab_nt_height_factors.txt
