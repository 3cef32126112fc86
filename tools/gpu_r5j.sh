#!/bin/bash
# round 5, visit j: wave-priority experiment of the role-split dK / dV kernel (config 4, same box)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "role_split or attention_bwd" > $OUT/r5j_pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/r5j_pytest.log | cut -c1-200
AB_ARGS="--config 4" bash tools/ab5.sh 2 "c4 role split, no priorities|" "c4 S-waves prio 1|UVTG_ATTN_WS_PRIO=1" "c4 P-waves prio 1|UVTG_ATTN_WS_PRIO=2" "c4 S-waves prio 1 during their MFMAs|UVTG_ATTN_WS_PRIO=3" 2>&1 | tee $OUT/r5j_ab.txt
