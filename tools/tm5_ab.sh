#!/bin/bash
# 320-row tiles of the persistent NT GEMM: GPU suite, then whole-step A/B in one box (UVTG_NT_TM5_OFF=1 = heights <= 256 only)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-padded-compare ${CFG:-} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['all_gemm_kernels']['gemm_nt256_kernel'])"; }
for rep in 1 2; do
  echo "tm<=4  $(UVTG_NT_TM5_OFF=1 run)"
  echo "tm5    $(run)"
done
for c in 3 5; do
  echo "config $c tm<=4  $(CFG="--config $c" UVTG_NT_TM5_OFF=1 run)"
  echo "config $c tm5    $(CFG="--config $c" run)"
done
