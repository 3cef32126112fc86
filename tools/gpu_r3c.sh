#!/bin/bash
# round-3 visit C: suite, A/B of the fused head pass and of the batched conv / projection weight gradients, stride probe, per-step kernel table
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp; TAG=${1:-r03c}
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|Error" $OUT/${TAG}_pytest_gpu.log | head -20
bash tools/ab_env.sh "UVTG_HEADFUSE_OFF=1" "" 2>&1 | tee $OUT/${TAG}_ab_headfuse.log
bash tools/ab_env.sh "UVTG_TN_CONVBATCH_OFF=1 UVTG_TN_PROJBATCH_OFF=1" "" 2>&1 | tee $OUT/${TAG}_ab_tnbatch.log
timeout 300 python tools/stride_probe.py 2>&1 | tee $OUT/${TAG}_stride_probe.log
bash tools/prof.sh ${TAG}c2 26 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
head -70 $OUT/${TAG}c2_stats.md | cut -c1-200
