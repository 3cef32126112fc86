#!/bin/bash
# round-3 visit D: suite, head-fusion A/B, per-step kernel table
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp; TAG=${1:-r03d}
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|Error" $OUT/${TAG}_pytest_gpu.log | head -20
bash tools/ab_env.sh "UVTG_HEADFUSE_OFF=1" "" 2>&1 | tee $OUT/${TAG}_ab_headfuse.log
bash tools/prof.sh ${TAG}c2 26 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
head -60 $OUT/${TAG}c2_stats.md | cut -c1-150 | grep -v "at6native"
