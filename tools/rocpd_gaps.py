#!/usr/bin/env python3
"""GPU idle time between kernels from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME -- cmd):
over the last FRAC of the trace (default 0.5: the timed steps, not the set-up) the span, the union of kernel intervals and
the idle remainder, plus the histogram of the gaps.  Usage: tools/rocpd_gaps.py DB [frac]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = db.execute("select start, end from rocpd_kernel_dispatch order by start").fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t1 - frac * (t1 - t0)
rows = [r for r in rows if r[0] >= cut]
span = max(r[1] for r in rows) - rows[0][0]
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
for s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"{len(rows)} dispatches over {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms ({100 * (span - busy) / span:.1f} %)")
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 1e4), (1e4, 2e4), (2e4, 1e5), (1e5, 1e12)):
    g = [x for x in gaps if lo <= x < hi]
    print(f"  gaps {lo / 1e3:6.0f}..{hi / 1e3:<9.0f} us: {len(g):5d}  total {sum(g) / 1e6:7.3f} ms")
