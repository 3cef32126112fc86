#!/bin/bash
# round 4, visit i: rehearsal of the driver's round-end commands on a fresh box (GPU suite, smoke, the literal bench command)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2>$OUT/bench_err.log; echo "bench rc=$?"; cut -c1-330 $OUT/bench_driver_cmd.json
python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "bench_driver_cmd.json")).read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "median", d["ms_per_step_event_median"], "over", d["median_over_steps"], "| roofline", r["frac"], r["traffic"], r["traffic_over_algorithmic"],
      "| encoder", d["roofline_encoder"]["frac"], "| cpu", d["cpu_baseline"]["value"], "| companions", {k: v["ms_per_step"] for k, v in d["companions"].items()})
PY
