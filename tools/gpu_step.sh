#!/bin/bash
# kernel + model GPU tests, then the config-2 bench line and a per-kernel profile of the step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-padded-compare --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
bash tools/prof.sh cur 13 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
grep -n "reduce\|total kernel" $OUT/cur_stats.md | cut -c1-140
