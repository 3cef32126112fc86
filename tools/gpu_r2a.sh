#!/bin/bash
# GPU visit: new full-size parity tests (all failures reported), the rest of the GPU suite, bench lines.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity_full.py -m gpu -q -s -rA 2>&1 | tail -150 ) > $OUT/parity_full.log
( timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -30 ) > $OUT/pytest_rest.log
( timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -3 ) > $OUT/bench_c2.log
tail -60 $OUT/parity_full.log; tail -12 $OUT/pytest_rest.log; cat $OUT/bench_c2.log
