#!/bin/bash
# round 4, visit r: FINAL build -- whole GPU suite, smoke, the single-tile variants' traces and A/Bs again (final tile shapes), then the artifact set
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
{
UVTG_NT_SMALL_OFF=1 timeout 200 python tools/nt_trace_infer.py 32 2>&1 | tail -25
timeout 200 python tools/nt_trace_infer.py 32 2>&1 | tail -25
timeout 200 python tools/nt_trace_infer.py 1 2>&1 | tail -25
timeout 200 python tools/nt_trace_infer.py 32 bf16 2>&1 | tail -25
} > $OUT/r04_nt_small_tile_phases.txt 2>&1
{
for i in 1 2; do
echo "persistent kernel only (UVTG_NT_SMALL_OFF=1)         : $(UVTG_NT_SMALL_OFF=1 timeout 300 python tools/infer_ab.py 2>&1 | tail -1)"
echo "single-tile variant, 128 x 256 tiles, no K split     : $(UVTG_NT_SMALL_TM1_OFF=1 UVTG_NT_SPLITK_MAX=0 timeout 300 python tools/infer_ab.py 2>&1 | tail -1)"
echo "single-tile variant, 128 x 128 / 256, no K split     : $(UVTG_NT_SPLITK_MAX=0 timeout 300 python tools/infer_ab.py 2>&1 | tail -1)"
echo "default (128 x 128 / 256 tiles, K split by cost model): $(timeout 300 python tools/infer_ab.py 2>&1 | tail -1)"
done
} | tee $OUT/r04_ab_nt_small_launches.txt
bash tools/ab_env.sh "UVTG_NT_SMALL_OFF=1" "" --no-companions 2>&1 | tee $OUT/r04_ab_nt_small_train.txt
timeout 300 python tools/batch_indep.py 2>&1 | tail -4 | tee $OUT/r04_batch_independence.txt
bash tools/gpu_r4q.sh
