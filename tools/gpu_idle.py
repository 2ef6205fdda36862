#!/usr/bin/env python
"""GPU idle time inside the steady-state steps of a rocprofv3 --kernel-trace run of bench.py (default two streams):
union of all kernel intervals over the last `frac` of the trace against the wall span, and the largest gaps with the
kernels on either side.   Usage: gpu_idle.py results.db [frac=0.5]"""
import sqlite3
import sys


def main(path, frac=0.5):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t1 - (t1 - t0) * frac
    rows = [r for r in rows if r[0] >= lo]
    busy, cur_end, gaps = 0, rows[0][0], []
    last_name = rows[0][2]
    for st, en, name in rows:
        if st > cur_end:
            gaps.append((st - cur_end, last_name, name))
            busy += en - st
            cur_end = en
        elif en > cur_end:
            busy += en - cur_end
            cur_end = en
        if en >= cur_end:
            last_name = name
    span = cur_end - rows[0][0]
    print(f"# window {span / 1e6:.2f} ms, {len(rows)} dispatches: GPU busy {busy / 1e6:.2f} ms = {100 * busy / span:.1f} %, idle {100 - 100 * busy / span:.1f} % "
          f"in {len(gaps)} gaps (median {sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3:.1f} us)")
    for g, a, b in sorted(gaps, reverse=True)[:12]:
        print(f"  {g / 1e3:8.1f} us between {a[:70]} -> {b[:70]}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
