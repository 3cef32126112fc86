#!/bin/bash
# round 4, visit k: epilogue-operand prefetch (groups at 256 rows, two-piece ring at 320 rows / gather) -- kernel + tile-equivalence tests, then
# in-box A/B against the previous behaviour (univtg_amd/libuvtg_prev.so = the same sources with -DUVTG_NT_GROUPS_MAX_TM=3 -DUVTG_NT_EOP_RING=0),
# headline (variant A) and variant B, two alternating rounds each
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "nt256 or tile or linear or engine_path or native_train_step or losses_and_grads" 2>&1 | tail -3
ab() {
  for round in 1 2; do
  for lib in $R/univtg_amd/libuvtg_prev.so $R/univtg_amd/libuvtg.so; do
    UVTG_LIB_PATH=$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-companions "$@" 2>/dev/null | tail -1 > /tmp/b.json
    python - "$lib" "$*" <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read())
g = d['roofline']['all_gemm_kernels']
print(f"{sys.argv[2]:12s} {sys.argv[1].split('/')[-1]:18s} step {d['ms_per_step']:.3f} ms (median {d['ms_per_step_event_median']:.3f}) enc {d['t_encoder_ms']:.3f} fwd/bwd {d['sections']['forward_ms']:.3f}/{d['sections']['backward_ms']:.3f} | nt256 {g['gemm_nt256_kernel']['ms_per_step']:.3f} ms {g['gemm_nt256_kernel']['tflops']:.0f} TF")
PY
  done; done
}
( ab; ab --variant B ) | tee $OUT/r04_ab_epilogue_operand_prefetch.txt
