#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp; TAG=r03j
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|Error" $OUT/${TAG}_pytest_gpu.log | head -20
bash tools/ab_env.sh "UVTG_LN_WIDE_WAVE_OFF=1" "" 2>&1 | tee $OUT/${TAG}_ab_lnwide.log
bash tools/prof.sh ${TAG}c2 26 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
grep -n "ln_fwd_wide\|per step" $OUT/${TAG}c2_stats.md | cut -c1-160
