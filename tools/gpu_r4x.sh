#!/bin/bash
# round 4, visit x: what do power and clocks do while the train step runs?  rocm-smi samples beside a 600-step bench run, and beside the inference loop
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
{
echo "== idle"; rocm-smi --showpower --showclocks --showuse 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|busy\|use" | head -12
( python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-companions > $OUT/power_bench.json 2>/dev/null ) &
BP=$!
sleep 25
for i in 1 2 3 4 5 6 7 8; do echo "== train step running, sample $i"; rocm-smi --showpower --showclocks --showuse 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|busy\|use" | head -12; sleep 0.7; done
wait $BP
tail -1 $OUT/power_bench.json | cut -c1-160
rocm-smi --showmaxpower 2>&1 | grep -i "power" | head -4
} 2>&1 | tee $OUT/r04_power_clock_samples.txt
