#!/bin/bash
# round 5, visit d: full GPU suite on the new planner (QKV = 4 rounds of 320-row tiles + single-tile tail) and the deferred LayerNorm reduce; A/B of both
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/r5d_pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/r5d_pytest.log | cut -c1-300
bash tools/ab5.sh 2 "default (new plan + LN reduce deferred)|" "LN reduce per launch|UVTG_LN_DEFER_OFF=1" "QKV as before (192-row single)|UVTG_NT_PLAN_OVR=27392,3072,192,0,0" 2>&1 | tee $OUT/r5d_ab.txt
