#!/bin/bash
# round 4, visit t: loader-wave experiment (UVTG_NT_LW = bit mask over tile heights 128 / 192 / 256 / 320): tile-path parity, then the train step A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
UVTG_NT_LW=15 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "tile or linear or small" 2>&1 | tail -3
{
for i in 1 2; do
for m in 0 15 12 4 8 3; do
  echo "UVTG_NT_LW=$m UVTG_NT_ORD=$( [ $m = 0 ] && echo 0011 || echo 0000 ): $(UVTG_NT_LW=$m bash -c 'python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-companions 2>/dev/null | tail -1' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); g=d['roofline']['all_gemm_kernels']['gemm_nt256_kernel']; print(f\"step {d['ms_per_step']:.3f} (median {d['ms_per_step_event_median']:.3f}) enc {d['t_encoder_ms']:.3f} nt256 {g['ms_per_step']:.3f} ms {g['tflops']:.0f} TF\")")"
done
done
} | tee $OUT/r04_ab_nt_loader_waves.txt
