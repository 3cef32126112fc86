#!/bin/bash
# round 5, visit e: lean LayerNorm forward -- kernel test, model tests that run it at production width, in-box A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "layernorm" > $OUT/r5e_pytest_ln.log 2>&1; echo "pytest ln rc=$?"
tail -3 $OUT/r5e_pytest_ln.log | cut -c1-300
grep -n "^E " $OUT/r5e_pytest_ln.log | head
timeout 1500 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_model.py -m gpu -q -x > $OUT/r5e_pytest_model.log 2>&1; echo "pytest model rc=$?"
tail -3 $OUT/r5e_pytest_model.log | cut -c1-300
bash tools/ab5.sh 2 "default (lean LN forward)|" "generic LN forward|UVTG_LN_FWD_LEAN_OFF=1" 2>&1 | tee $OUT/r5e_ab.txt
