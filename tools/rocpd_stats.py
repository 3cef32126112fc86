#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite result (--kernel-trace) into the per-kernel table `--stats` prints:
name, calls, total / average / min / max duration, share of GPU kernel time.  Usage:
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [steps] > profiles/rNN_kernel_stats.md
`steps` (optional) divides the per-step columns (total dispatches include warm-up steps)."""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = c.execute(f"select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
                     f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size) "
                     f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    span = c.execute(f"select min(start), max(end) from {kd}").fetchone()
    print(f"# rocprofv3 --kernel-trace summary of {db.split('/')[-1]}")
    print(f"\ntotal kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches; first-to-last span {(span[1] - span[0]) / 1e6:.3f} ms"
          + (f"; {steps} steps traced -> {tot / 1e6 / steps:.3f} ms kernel time per step" if steps else ""))
    # idle time between consecutive dispatches INSIDE a training step (a step ends with adamw_kernel): what a graph launch could remove
    seq = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    ends = [i for i, r in enumerate(seq) if "adamw_kernel" in (r[0] or "")]
    if len(ends) >= 3:
        gaps, busy, wall, n = [], [], [], []
        for a, b in zip(ends[1:-1], ends[2:]):
            st = seq[a + 1: b + 1]
            gaps.append(sum(max(0, st[i + 1][1] - st[i][2]) for i in range(len(st) - 1)))
            busy.append(sum(r[2] - r[1] for r in st)); wall.append(st[-1][2] - st[0][1]); n.append(len(st))
        k = len(gaps)
        print(f"\nper step (median of {k}): {sorted(n)[k // 2]} dispatches, first-start to last-end {sorted(wall)[k // 2] / 1e6:.3f} ms, kernel time "
              f"{sorted(busy)[k // 2] / 1e6:.3f} ms, idle between dispatches {sorted(gaps)[k // 2] / 1e6:.3f} ms "
              f"({sorted(gaps)[k // 2] / max(sorted(n)[k // 2] - 1, 1) / 1e3:.2f} us per gap)")
    print("\n| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | scratch B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for n, cnt, t, mn, mx, vg, ag, sg, lds, scr in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n or "?")
        n = n if len(n) < 90 else n[:87] + "..."
        print(f"| `{n}` | {cnt} | {t / 1e6:.3f} | {t / cnt / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * t / tot:.1f} | {vg} | {ag} | {sg} | {lds} | {scr} |")


if __name__ == "__main__":
    main()
