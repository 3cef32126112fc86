#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OLD=${1:-$R/tools/libuvtg_old.so}
UVTG_LIB_PATH=$OLD bash tools/prof.sh old 13 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
bash tools/prof.sh new 13 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
for f in old new; do echo "== $f"; grep "gemm\|total kernel" gpurun_out/${f}_stats.md | cut -c1-170; done
