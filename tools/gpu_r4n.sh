#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 300 python tools/batch_indep.py 2>&1 | tail -6
timeout 300 bash tools/prof.sh r04inf32sk 40 python $R/tools/infer_prof.py 32 > /dev/null 2>&1; head -16 $OUT/r04inf32sk_stats.md | cut -c1-150
timeout 300 bash tools/prof.sh r04inf1sk 40 python $R/tools/infer_prof.py 1 > /dev/null 2>&1; head -16 $OUT/r04inf1sk_stats.md | cut -c1-150
