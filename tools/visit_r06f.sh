#!/bin/bash
# round-6 visit f: the TAL branch (src_cls + 'saliency_cls') on the GPU, then the whole suite
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "tiny_tal or tal_branch" > $OUT/r06f_pytest_tal.log 2>&1; echo "pytest tal rc=$?"; grep -n "passed\|failed" $OUT/r06f_pytest_tal.log | tail -1
grep -n "^FAILED\|^E  " $OUT/r06f_pytest_tal.log | head -30
timeout 2400 python -m pytest tests -m gpu -q > $OUT/r06f_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/r06f_pytest_gpu.log | tail -1
grep -n "^FAILED" $OUT/r06f_pytest_gpu.log | head -20
