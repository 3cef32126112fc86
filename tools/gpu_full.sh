#!/bin/bash
# Full GPU visit: whole GPU suite, smoke, bench lines for every config
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "^\[\|passed\|failed\|^E  \|Error" $OUT/pytest_gpu.log | head -40
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $OUT/smoke.log; cat $OUT/smoke.log
( timeout 600 python bench.py 2>/dev/null | tail -1 ) > $OUT/bench_c2.json; cut -c1-300 $OUT/bench_c2.json
for c in 3 4 5; do ( timeout 300 python bench.py --config $c --steps 20 --warmup 5 2>/dev/null | tail -1 ) > $OUT/bench_c$c.json; python - $OUT/bench_c$c.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(d['config']['baseline_config'], d['value'], d['ms_per_step'], d['t_encoder_ms'], d['roofline_encoder']['frac'], d['padded_execution_ms_per_step'], d['encoder_rows_fraction'])
PY
done
