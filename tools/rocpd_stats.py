#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table:
calls, total ms, average us, % of GPU kernel time, grid, VGPR/LDS.  Usage: rocpd_stats.py results.db [skip_first_n_ms]"""
import sqlite3
import sys
from collections import defaultdict


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, "
                       f"d.workgroup_size_x, s.arch_vgpr_count, s.accum_vgpr_count, d.group_segment_size "
                       f"from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
    agg = defaultdict(lambda: [0, 0.0, None])
    tot = 0.0
    for name, st, en, gx, gy, gz, wx, vg, ag, lds in rows:
        dt = (en - st) / 1e3
        key = (name, gx // max(wx, 1), gy, gz)
        a = agg[key]
        a[0] += 1
        a[1] += dt
        a[2] = (vg, ag, lds)
        tot += dt
    print(f"# {path}: {len(rows)} dispatches, total kernel time {tot / 1e3:.3f} ms")
    print(f"{'%':>6} {'total_ms':>10} {'calls':>6} {'avg_us':>10}  {'blocks(x,y,z)':>18} {'vgpr/agpr/lds':>16}  kernel")
    for (name, bx, by, bz), (n, t, meta) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"{100 * t / tot:6.2f} {t / 1e3:10.3f} {n:6d} {t / n:10.1f}  {str((bx, by, bz)):>18} {str(meta):>16}  {short}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
