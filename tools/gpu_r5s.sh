#!/bin/bash
# round 5, visit s: full GPU suite on the last build (graph LRU, merged frame zeroing, two blocks per sample in the head pass), A/B of the head pass,
# per-step dispatch count, and the tile-shape ceiling probe of the dominant kernel
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $OUT/r5s_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5s_pytest.log | cut -c1-200; grep -n "FAILED\|^E  " $OUT/r5s_pytest.log | head
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 )
bash tools/ab5.sh 2 "head pass: one block per sample|UVTG_HEADFUSE_SPLIT=1" "head pass: two blocks per sample (default)|" "head pass: four blocks per sample|UVTG_HEADFUSE_SPLIT=4" 2>&1 | tee $OUT/r5s_ab.txt
python tools/nt_ceiling.py 2>/dev/null | tee $OUT/r5s_nt_ceiling.txt
timeout 400 bash tools/prof.sh r5sc2 23 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-companions > /dev/null 2>&1
sed -n 3,6p $OUT/r5sc2_stats.md
