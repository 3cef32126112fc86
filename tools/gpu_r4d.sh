#!/bin/bash
# round 4, visit d: whole GPU suite (graph replay test, split kernel at nt256 shapes, txt_pos train replay), inference line with HIP-graph cases,
# per-kernel table of the B=32 precise inference call
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|^FAILED\|^\[config2 fp32\|^\[seeds\|   sample\|^\[linear_f32x3" $OUT/pytest_gpu.log | cut -c1-220 | head -50
( timeout 400 python bench.py --mode infer 2>$OUT/infer_err.log | tail -1 ) > $OUT/r04_bench_infer.json; cut -c1-200 $OUT/r04_bench_infer.json; tail -3 $OUT/infer_err.log
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r04_bench_infer.json")))
print({k: v["ms_per_batch"] for k, v in d["cases"].items()})
PY
timeout 300 bash tools/prof.sh r04inf32 40 python $R/tools/infer_prof.py 32 > /dev/null 2>&1; head -45 $OUT/r04inf32_stats.md | cut -c1-170
