#!/usr/bin/env python3
"""Kernel-trace timeline digest of a rocpd database: per-kernel totals, the wall span, the sum of durations and
the average number of kernels executing at once (steady-state part: the middle half of the trace)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
t0, t1 = rows[0][1], max(r[2] for r in rows)
lo, hi = t0 + (t1 - t0) // 4, t0 + 3 * (t1 - t0) // 4
mid = [r for r in rows if r[1] >= lo and r[2] <= hi]
span = hi - lo
by = {}
for name, s, e in mid:
    k = name.split('(')[0].split('::')[-1][:40]
    a = by.setdefault(k, [0, 0])
    a[0] += 1; a[1] += e - s
print(f"kernels in window: {len(mid)}  window {span/1e6:.3f} ms  sum of durations {sum(e-s for _,s,e in mid)/1e6:.3f} ms  "
      f"avg concurrency {sum(e-s for _,s,e in mid)/span:.2f}")
for k, (n, d) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:<42} calls {n:>6}  total {d/1e6:>9.3f} ms  avg {d/n/1e3:>8.2f} us  busy-share {d/span:>6.2f}")
