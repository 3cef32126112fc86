#!/bin/bash
# round 4, visit e: small-M launch heuristic A/B on the inference call (two rounds, alternating), then the kernel + model tests that cover it
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for r in 1 2; do
  echo "smallM off: $(UVTG_NT_SMALLM_OFF=1 timeout 200 python tools/infer_ab.py 2>/dev/null | tail -1)"
  echo "smallM on : $(timeout 200 python tools/infer_ab.py 2>/dev/null | tail -1)"
done | tee $OUT/r04_ab_small_m_heuristic.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -3
