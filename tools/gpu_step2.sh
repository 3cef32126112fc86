#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -2
for i in 1 2; do python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-padded-compare --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"; done
timeout 300 python tools/nt_trace.py 2 > $OUT/nt_trace_c2.txt 2>&1; sed -n 2,8p $OUT/nt_trace_c2.txt; sed -n 24,30p $OUT/nt_trace_c2.txt
