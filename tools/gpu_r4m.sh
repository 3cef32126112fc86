#!/bin/bash
# round 4, visit m: split-K of the small NT launches -- kernel test, the inference parity tests, then the in-box A/B over the part cap
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "split_k or tile or linear" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -8
{
for round in 1 2; do
  for cap in 0 2 3 4 6 8; do
    echo "UVTG_NT_SPLITK_MAX=$cap: $(UVTG_NT_SPLITK_MAX=$cap timeout 300 python tools/infer_ab.py 2>&1 | tail -1)"
  done
done
} | tee $OUT/r04_ab_nt_split_k.txt
