#!/bin/bash
# round 5, visit l: head-major block decode of the tiled attention kernels (XCD balance on ragged batches) -- attention tests, config-4/5 tests, A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" > $OUT/r5l_pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/r5l_pytest.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q -x -k "config4 or config5 or config3 or dropout" > $OUT/r5l_pytest_c4.log 2>&1; echo "pytest configs rc=$?"; tail -1 $OUT/r5l_pytest_c4.log | cut -c1-200
AB_ARGS="--config 4" bash tools/ab5.sh 2 "c4 sample-major decode (as before)|UVTG_ATTN_SAMPLE_MAJOR=1" "c4 head-major decode (default)|" 2>&1 | tee $OUT/r5l_ab.txt
AB_ARGS="--config 5" bash tools/ab5.sh 2 "c5 sample-major decode (as before)|UVTG_ATTN_SAMPLE_MAJOR=1" "c5 head-major decode (default)|" 2>&1 | tee -a $OUT/r5l_ab.txt
AB_ARGS="--config 3" bash tools/ab5.sh 1 "c3 sample-major decode (as before)|UVTG_ATTN_SAMPLE_MAJOR=1" "c3 head-major decode (default)|" 2>&1 | tee -a $OUT/r5l_ab.txt
