#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x > $OUT/pytest_a.log 2>&1; tail -15 $OUT/pytest_a.log
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $OUT/bench_c2.log
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/bench_c2.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','t_encoder_ms','sections','padded_execution_ms_per_step')}, d['roofline_encoder']['frac'])
r=d['roofline']; print(r['achieved'], r['avg_launch_us'], r['all_gemm_kernels'])
PY
