#!/bin/bash
# round 5, visit f: delta fused into the dO GEMM epilogue (EPI 4) -- full GPU suite, then A/B (delta fusion, LN forward block count)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x > $OUT/r5f_pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/r5f_pytest.log | cut -c1-300
grep -n "^E  \|FAILED" $OUT/r5f_pytest.log | head -20
bash tools/ab5.sh 2 "default (delta in the dO GEMM epilogue)|" "attn_delta_kernel pass|UVTG_DELTA_FUSE_OFF=1" "LN fwd 512 blocks|UVTG_LN_FWD_BLOCKS=512" "LN fwd 2048 blocks|UVTG_LN_FWD_BLOCKS=2048" 2>&1 | tee $OUT/r5f_ab.txt
AB_ARGS="--variant B" bash tools/ab5.sh 1 "variant B default|" "variant B attn_delta_kernel pass|UVTG_DELTA_FUSE_OFF=1" 2>&1 | tee -a $OUT/r5f_ab.txt
