#!/bin/bash
# round 5, visit a: the evidence gaps -- headline workload replayed through the oracle, TrainStep layout-flip test, the default bench line with
# the REAL reference as cpu_baseline (oracle/_ref archive) and every other BASELINE config as companions
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
ls -la oracle/_ref/ > $OUT/r5a_ref_ls.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity_full.py -m gpu -q -s -k "bench_path_trainstep" > $OUT/r5a_pytest_replay.log 2>&1; echo "replay rc=$?"
tail -5 $OUT/r5a_pytest_replay.log
grep -n "config2-" $OUT/r5a_pytest_replay.log | head
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "projection_mode_flip" > $OUT/r5a_pytest_flip.log 2>&1; echo "flip rc=$?"
tail -3 $OUT/r5a_pytest_flip.log
( timeout 900 python bench.py 2>$OUT/r5a_bench.err | tail -1 ) > $OUT/r5a_bench.json; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5a_bench.json').read())
print("ms/step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], "enc", d.get("roofline_encoder"))
c = d["cpu_baseline"]; print("cpu:", c["kind"], c["value"], c.get("port_over_reference"), c.get("port_beside_it", {}).get("value"))
for k, v in (d.get("companions") or {}).items():
    if isinstance(v, dict):
        print(k, {kk: v[kk] for kk in ("ms_per_step", "valid_clips_per_sec", "t_encoder_ms") if kk in v}, v.get("roofline_encoder"), {kk: vv for kk, vv in v.items() if kk.startswith("config")})
PY
