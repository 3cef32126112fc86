#!/bin/bash
# round 5, visit c: NT launch-plan experiments, in-box A/B on alternating bench runs (headline workload, variant A):
#   (1) column-group tile order of the N = 3072 launches (UVTG_NT_CGW = 6 / 4)   (2) QKV as 4 rounds of 320-row tiles + a 192-row tail
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "column_group or tile_heights or loader_waves_bit or nt_small" > $OUT/r5c_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/r5c_pytest.log
run() {   # label, env...
  local L="$1"; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-companions 2>/dev/null | tail -1 > /tmp/b.json
  python - "$L" <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read())
g = d['roofline']['all_gemm_kernels']
print(f"{sys.argv[1]:44s} step {d['ms_per_step']:.3f} ms (median {d['ms_per_step_event_median']:.3f}) enc {d['t_encoder_ms']:.3f} fwd/bwd {d['sections']['forward_ms']:.3f}/{d['sections']['backward_ms']:.3f} | nt256 {g['gemm_nt256_kernel']['ms_per_step']:.3f} ms {g['gemm_nt256_kernel']['tflops']:.0f} TF ({g['gemm_nt256_kernel']['launches_per_step']})")
PY
}
for round in 1 2; do
  run "default" X=1
  run "UVTG_NT_CGW=6" UVTG_NT_CGW=6
  run "UVTG_NT_CGW=4" UVTG_NT_CGW=4
  run "QKV 320x(27200)+192 tail" "UVTG_NT_PLAN_OVR=27392,3072,320,27200,192"
  run "QKV 320x(27200)+128 tail (single-tile)" "UVTG_NT_PLAN_OVR=27392,3072,320,27200,128"
  run "QKV 256x(27136)+128 tail (single-tile)" "UVTG_NT_PLAN_OVR=27392,3072,256,27136,128"
  run "QKV 320 single" "UVTG_NT_PLAN_OVR=27392,3072,320,0,0"
  run "QKV 320+128 tail, CGW=6" UVTG_NT_CGW=6 "UVTG_NT_PLAN_OVR=27392,3072,320,27200,128"
done 2>&1 | tee $OUT/r5c_ab_nt_plans.txt
