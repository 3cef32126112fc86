#!/bin/bash
# round 5, visit o: dQ kernel without the vmcnt(0) behind its DMA issue -- tests, attn_bench (kernel stats), config 4 / 5 bench lines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" > $OUT/r5o_pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/r5o_pytest.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q -x -k "config4 or config5 or dropout" > $OUT/r5o_pytest_c4.log 2>&1; echo "pytest configs rc=$?"; tail -1 $OUT/r5o_pytest_c4.log | cut -c1-200
python tools/attn_bench.py 2>/dev/null | tee $OUT/r5o_attn_bench.txt
cd /tmp; rm -rf /tmp/st2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st2 -o ab -- python $R/tools/attn_bench.py > /dev/null 2>&1
cut -d, -f1-4 $(find /tmp/st2 -name '*kernel_stats.csv' | head -1) | head -7 | tee $OUT/r5o_attn_bench_kernel_stats.csv
cd $R
AB_ARGS="--config 4" bash tools/ab5.sh 2 "c4|" 2>&1 | tee $OUT/r5o_ab.txt
AB_ARGS="--config 5" bash tools/ab5.sh 1 "c5|" 2>&1 | tee -a $OUT/r5o_ab.txt
