#!/bin/bash
# round 5, visit b: per-dispatch sequence of one headline step (rocprofv3 kernel trace), the flip test again, and the LayerNorm-forward
# event anomaly of visit a (is it the companions' presence?)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "projection_mode_flip" > $OUT/r5b_pytest_flip.log 2>&1; echo "flip rc=$?"
tail -3 $OUT/r5b_pytest_flip.log
( timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 ) > $OUT/r5b_bench_nocomp.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5b_bench_nocomp.json').read())
print("no-other-configs: ms/step", d["ms_per_step"], "LN fwd", d["roofline_hbm"]["layernorm_forward"]["ms_per_step"], "LN bwd", d["roofline_hbm"]["layernorm_backward"]["ms_per_step"])
PY
cd /tmp
rm -rf /tmp/p_seq
rocprofv3 --kernel-trace -d /tmp/p_seq -o seq -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-companions --profile-steps 0 > /dev/null 2>&1
DB=$(find /tmp/p_seq -name '*.db' | head -1)
python $R/tools/rocpd_seq.py $DB 8 > $OUT/r5b_seq_config2.txt
python $R/tools/rocpd_stats.py $DB 12 > $OUT/r5b_stats_config2.md
wc -l $OUT/r5b_seq_config2.txt; head -5 $OUT/r5b_stats_config2.md
