#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "loader_waves" 2>&1 | grep -v "^$" | tail -25
