#!/bin/bash
# usage: tools/prof.sh <name> <steps-for-per-step-column> <command...>   -> gpurun_out/<name>_stats.md
R=${GRAFT_REPO_ROOT:-/root/repo}; N=$1; S=$2; shift 2
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$N
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$N -o $N -- "$@" > $R/gpurun_out/prof_$N.log 2>&1
DB=$(find $R/gpurun_out/prof_$N -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB $S > $R/gpurun_out/${N}_stats.md
rm -rf $R/gpurun_out/prof_$N
head -60 $R/gpurun_out/${N}_stats.md
