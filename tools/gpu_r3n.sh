#!/bin/bash
# visit N: software-pipelined dK/dV kernel (scores of block qb + products of block qb-1 per iteration, fenced interleave)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
PREV=$R/univtg_amd/libuvtg_prev.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "attention" 2>&1 | tail -3
echo "== attn_bench: new"; timeout 200 python tools/attn_bench.py 2>&1 | grep "^B=" | tee $OUT/r03n_attn_new.txt
echo "== attn_bench: previous build"; UVTG_LIB_PATH=$PREV timeout 200 python tools/attn_bench.py 2>&1 | grep "^B=" | tee $OUT/r03n_attn_prev.txt
echo "== per-kernel durations (rocprofv3 --kernel-trace --stats over tools/attn_bench.py)"
( cd /tmp && rm -rf /tmp/attnprof && rocprofv3 --kernel-trace --stats -d /tmp/attnprof -o a --output-format csv -- python $R/tools/attn_bench.py > /dev/null 2>&1 )
F=$(find /tmp/attnprof -name '*kernel_stats.csv' | head -1)
python - "$F" <<'PY' | tee $OUT/r03n_attn_kernel_stats.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "attn" in n: print(f"{n[:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f}")
PY
line() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ra = d.get("roofline_attention", {})
print(f"   {d['ms_per_step']:.3f} ms/step  t_encoder {d.get('t_encoder_ms')}  attn fwd {ra.get('forward', {}).get('ms_per_step')} bwd {ra.get('backward', {}).get('ms_per_step')}")
PY
}
for cfg in 4 2; do
  for arm in new prev; do
    unset UVTG_LIB_PATH
    if [ $arm = prev ]; then export UVTG_LIB_PATH=$PREV; fi
    timeout 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-padded-compare 2>/dev/null | tail -1 > /tmp/b.json
    echo "config $cfg $arm:"; line /tmp/b.json
  done
done 2>&1 | tee $OUT/r03n_step_ab.txt
unset UVTG_LIB_PATH
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x > $OUT/r03n_pytest_model.log 2>&1; echo "model tests rc=$?"; grep -n "passed\|failed" $OUT/r03n_pytest_model.log | tail -2
