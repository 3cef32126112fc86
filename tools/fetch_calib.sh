#!/bin/bash
# FETCH_SIZE calibration for the NT GEMM's staging pattern: rocprofv3 --pmc FETCH_SIZE over tools/lab/l2_lds_bw calib, which
# reads 128 MiB exactly once with (1) the kernel's LDS-DMA pattern (8 rows x 128 B per wave-instruction), (2) the same addresses into
# registers, (3) a linear 16 B/lane stream.  factor = known bytes / (FETCH_SIZE KiB x 1024).  -> gpurun_out/fetch_calibration.json
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/fcal
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d /tmp/fcal_$C -o p -- $R/tools/lab/l2_lds_bw calib > $OUT/fetch_calib_$C.log 2>&1
done
python - <<'PY'
import csv, glob, json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
known = 256 * 16 * 4 * 8 * 1024
out = {"known_bytes_per_kernel": known, "command": "tools/fetch_calib.sh: rocprofv3 --pmc FETCH_SIZE -- tools/lab/l2_lds_bw calib (one launch per pattern, every byte read once)", "patterns": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/fcal_{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"]
        name = "nt_staging_lds_dma" if "true" in k and "stream_kernel" in k else ("nt_staging_to_vgpr" if "stream_kernel" in k else ("linear_16B_per_lane" if "calib_linear" in k else None))
        if name is None:
            continue
        e = out["patterns"].setdefault(name, {})
        e[c + "_KiB"] = float(r["Counter_Value"])
        if c == "FETCH_SIZE":
            e["factor"] = round(known / (float(r["Counter_Value"]) * 1024), 4)
json.dump(out, open(f"{R}/gpurun_out/fetch_calibration.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
