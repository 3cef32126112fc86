#!/bin/bash
# round 4, visit h: the index-parity tests with the reference-self-consistency check; MFMA-busy / wait split of the split-operand instantiations
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_full.py -m gpu -q -s -k "post_nms or fp32x3" > $OUT/pytest_idx.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|^FAILED\|^\[config2 fp32\|^\[seeds\|   sample" $OUT/pytest_idx.log | cut -c1-220 | head -30
export PMC_EXTRA="--kernel-include-regex gemm_nt256_kernel<[^>]*true>"
bash tools/pmc.sh split "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" -- python $R/tools/infer_prof.py 256 > /dev/null 2>&1
cp $OUT/pmc_split_0.txt $OUT/r04_pmc_split_gemm_sq.txt; cat $OUT/r04_pmc_split_gemm_sq.txt | head -40
