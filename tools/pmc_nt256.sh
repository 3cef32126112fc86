#!/bin/bash
# Three separate rocprofv3 --pmc passes over the bench command (kernel filter: the persistent NT GEMM), then the per-launch
# HBM traffic + MFMA-busy summary bench.py quotes as roofline.traffic.  Outputs: gpurun_out/pmc_nt_{0,1,2}.txt, gpurun_out/pmc_nt256.json
R=${GRAFT_REPO_ROOT:-/root/repo}
export PMC_EXTRA="--kernel-include-regex gemm_nt256_kernel<([^,]*,){5}.false,"      # the bf16 instantiations (sixth template argument HALF = false: persistent, loader-wave and single-tile instantiations)
bash $R/tools/pmc.sh nt "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" -- \
  python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-companions --profile-steps 0 > /dev/null
python - <<'PY'
import json, os, re
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
def parse(i):
    out, cur = {}, None
    for l in open(f"{R}/gpurun_out/pmc_nt_{i}.txt"):
        if not l.startswith("   "):
            cur = l.strip(); out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+) mean=\s*([\d.]+)", l)
            if m: out[cur][m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return out
def wsum(d, name):      # launch-count weighted mean over every nt256 instantiation
    n = sum(v[name][0] for v in d.values() if name in v)
    return sum(v[name][0] * v[name][1] for v in d.values() if name in v) / max(n, 1), n
p0, p1, p2 = parse(0), parse(1), parse(2)
fetch, n = wsum(p0, "FETCH_SIZE"); gui, _ = wsum(p0, "GRBM_GUI_ACTIVE"); write, _ = wsum(p1, "WRITE_SIZE"); mfma, _ = wsum(p2, "SQ_VALU_MFMA_BUSY_CYCLES")
import sys
sys.path.insert(0, R)
from bench import kernel_src_sha
factor, fsrc = 2.0, "guide: wide coalesced streaming reads are tallied at half their bytes (MI355X_MICROARCH.md, HBM section); uncalibrated for this pattern"
for cal in (f"{R}/gpurun_out/fetch_calibration.json", f"{R}/profiles/r04_fetch_calibration.json", f"{R}/profiles/r03_fetch_calibration.json"):
    if os.path.exists(cal):
        cj = json.load(open(cal))
        pat = cj.get("patterns", {}).get("nt_staging_lds_dma", {})
        if "factor" in pat:
            factor, fsrc = pat["factor"], f"calibrated: tools/fetch_calib.sh, the kernel's own LDS-DMA staging pattern over a known byte count ({os.path.basename(cal)})"
            break
head = open(f"{R}/tools/.git_head").read().strip() if os.path.exists(f"{R}/tools/.git_head") else "unknown"
fb, wb = fetch * 1024 * factor, write * 1024
js = {"kernel_src_sha": kernel_src_sha(), "git_head": head, "config": 2, "variant": "A", "fetch_factor": factor, "fetch_factor_source": fsrc, "kernel": "gemm_nt256_kernel<*> (all instantiations, launch-weighted)",
      "command": "tools/pmc_nt256.sh: rocprofv3 --pmc <pass> --kernel-include-regex 'gemm_nt256_kernel<([^,]*,){5}.false,' -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-companions --profile-steps 0  (three separate passes: FETCH_SIZE GRBM_GUI_ACTIVE | WRITE_SIZE | SQ_*)",
      "launches_averaged": n, "per_instantiation": {k: {c: v[1] for c, v in d.items()} for k, d in {**p0}.items()},
      "FETCH_SIZE_KB_raw": round(fetch, 1), "WRITE_SIZE_KB_raw": round(write, 1),
      "fetch_bytes_corrected": int(fb), "write_bytes": int(wb), "traffic_bytes_per_launch": int(fb + wb),
      "correction": "FETCH_SIZE (KiB) x fetch_factor; WRITE_SIZE (KiB) taken as is",
      "SQ_VALU_MFMA_BUSY_CYCLES": mfma, "GRBM_GUI_ACTIVE_sum_over_8_xcd": gui,
      "mfma_busy_frac": round(mfma / (gui / 8 * 1024), 3),
      "sq_wave_cycle_split": {k: round(wsum(p2, k)[0] / max(wsum(p2, "SQ_WAVE_CYCLES")[0], 1), 3) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}}
json.dump(js, open(f"{R}/gpurun_out/pmc_nt256.json", "w"), indent=1)
print(json.dumps({k: v for k, v in js.items() if k != "per_instantiation"}, indent=1))
PY
