#!/bin/bash
# round 4, visit j: per-tile phase trace of the persistent NT GEMM inside one training step, variant A (headline) and variant B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 300 python tools/nt_trace.py 2 --full > $OUT/r04_nt_tile_phases_variantA.txt 2>$OUT/trace_err.log; tail -3 $OUT/trace_err.log; head -60 $OUT/r04_nt_tile_phases_variantA.txt | cut -c1-230
timeout 300 python tools/nt_trace.py 2 > $OUT/r04_nt_tile_phases_variantB.txt 2>>$OUT/trace_err.log; head -8 $OUT/r04_nt_tile_phases_variantB.txt | cut -c1-200
