#!/bin/bash
# round 5, visit q: P-wave requests its sixteen hand-off operands up front -- tests, attn_bench, config 4 / 5
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" > $OUT/r5q_pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/r5q_pytest.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q -x -k "config4 or config5 or dropout" > $OUT/r5q_pytest_c4.log 2>&1; echo "pytest configs rc=$?"; tail -1 $OUT/r5q_pytest_c4.log | cut -c1-200
for round in 1 2; do
  echo -n "role split (default)        "; python tools/attn_bench.py 2>/dev/null | head -1
  echo -n "role split, key-group P     "; UVTG_ATTN_WS_KEYP=1 python tools/attn_bench.py 2>/dev/null | head -1
  echo -n "one wave per SIMD           "; UVTG_ATTN_WS_OFF=1 python tools/attn_bench.py 2>/dev/null | head -1
done | tee $OUT/r5q_attn_bench.txt
AB_ARGS="--config 4" bash tools/ab5.sh 2 "c4 one-wave-per-SIMD dK/dV|UVTG_ATTN_WS_OFF=1" "c4 role split (default)|" 2>&1 | tee $OUT/r5q_ab.txt
