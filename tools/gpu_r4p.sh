#!/bin/bash
# round 4, visit p: single-tile NT variants (64 / 128-row tiles, three-stage ring, K split by the cost model): kernel + model tests,
# tile-phase traces of the inference call, inference A/B, and the training headline with / without the variant (the 8192-row text GEMMs take it)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -3
{
UVTG_NT_SMALL_OFF=1 timeout 200 python tools/nt_trace_infer.py 32 2>&1 | tail -25
timeout 200 python tools/nt_trace_infer.py 32 2>&1 | tail -25
timeout 200 python tools/nt_trace_infer.py 1 2>&1 | tail -25
timeout 200 python tools/nt_trace_infer.py 32 bf16 2>&1 | tail -25
} > $OUT/r04_nt_small_tile_phases.txt 2>&1
{
for i in 1 2; do
echo "persistent kernel only (UVTG_NT_SMALL_OFF=1)        : $(UVTG_NT_SMALL_OFF=1 timeout 300 python tools/infer_ab.py 2>&1 | tail -1)"
echo "single-tile variant, 128-row tiles, no K split      : $(UVTG_NT_SMALL_TM1_OFF=1 UVTG_NT_SPLITK_MAX=0 timeout 300 python tools/infer_ab.py 2>&1 | tail -1)"
echo "single-tile variant, 64/128-row tiles, no K split   : $(UVTG_NT_SPLITK_MAX=0 timeout 300 python tools/infer_ab.py 2>&1 | tail -1)"
echo "default (64/128-row tiles, K split by cost model)   : $(timeout 300 python tools/infer_ab.py 2>&1 | tail -1)"
done
} | tee $OUT/r04_ab_nt_small_launches.txt
bash tools/ab_env.sh "UVTG_NT_SMALL_OFF=1" "" --no-companions 2>&1 | tee $OUT/r04_ab_nt_small_train.txt
