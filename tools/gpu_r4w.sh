#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
{ echo "== loader waves (default mask 7)"; timeout 300 python tools/nt_ktrace.py --full 2>&1 | tail -48; echo "== every wave stages for itself (mask 0)"; timeout 300 python tools/nt_ktrace.py --full 0 2>&1 | tail -48; } > $OUT/r04_nt_ktile_cycles.txt 2>&1
cut -c1-200 $OUT/r04_nt_ktile_cycles.txt
