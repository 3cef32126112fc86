#!/usr/bin/env python
"""Per-dispatch durations of ONE training step (the step after the k-th adamw_kernel) from a rocprofv3 rocpd database:
    python tools/rocpd_seq.py results.db [k] -> lines "idx short_kernel_name duration_us" (GEMM kernels only with --gemm)"""
import re, sqlite3, sys
db = sys.argv[1]; k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
seg = rows[ends[k - 1] + 1: ends[k] + 1]
for i, (n, s, e) in enumerate(seg):
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n).replace("EEv8GemmArgs.kd", "").replace(".kd", "")
    if "--gemm" in sys.argv and "gemm" not in n: continue
    print(i, n[:60], f"{(e - s) / 1e3:.1f}")
