#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "small or tile or linear" 2>&1 | tail -5
{
timeout 200 python tools/nt_trace_infer.py 32 2>&1 | tail -25
timeout 200 python tools/nt_trace_infer.py 1 2>&1 | tail -25 | head -9
} > $OUT/nt_trace_infer.txt 2>&1
cut -c1-260 $OUT/nt_trace_infer.txt
for i in 1 2; do
echo "small off        : $(UVTG_NT_SMALL_OFF=1 timeout 300 python tools/infer_ab.py 2>&1 | tail -1)"
echo "default          : $(timeout 300 python tools/infer_ab.py 2>&1 | tail -1)"
done
