#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace) as a per-kernel table:
   python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--by-grid] > profiles/xyz_kernel_stats.txt
(--by-grid: one line per kernel and grid size, e.g. the four A-Trous passes)"""
import sqlite3
import sys


def main(path, by_grid=False):
    c = sqlite3.connect(path)
    key = "name || ' grid ' || grid_x" if by_grid else "name"
    rows = c.execute(f"select {key}, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     f"max(vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x) from kernels group by {key} "
                     "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"{'kernel':<70} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'%':>6} {'vgpr':>5} {'sgpr':>5} {'lds':>6} {'wg':>5}")
    for name, calls, tot, avg, mn, mx, vg, sg, lds, wg in rows:
        base = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = (base.split("(")[0] + (name[name.rindex(" grid "):] if by_grid else ""))[-70:]
        print(f"{short:<70} {calls:>6} {tot / 1e6:>10.3f} {avg / 1e3:>10.2f} {mn / 1e3:>9.2f} {mx / 1e3:>10.2f} "
              f"{100.0 * tot / total:>6.2f} {vg:>5} {sg:>5} {lds:>6} {wg:>5}")


def tail(path, spec):
    """--tail SUBSTRING:N  average duration of the LAST N dispatches of the kernels whose name contains SUBSTRING
    (bench.py's timed region is the end of a --no-extras run: its HIP-event figure roofline.avg_launch_us must agree)"""
    sub, n = spec.rsplit(":", 1)
    c = sqlite3.connect(path)
    rows = c.execute("select duration from kernels where name like ? order by start desc limit ?", (f"%{sub}%", int(n))).fetchall()
    if rows:
        d = [r[0] for r in rows]
        print(f"# last {len(d)} dispatches of *{sub}*: avg {sum(d) / len(d) / 1e3:.2f} us, min {min(d) / 1e3:.2f}, max {max(d) / 1e3:.2f}")


def timeline(path, n):
    """--timeline N  the last N dispatches of this library's kernels in start order: start (us after the first of them), duration, gap to the end of
    the latest dispatch before it (what a stream loses between dependent kernels)"""
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels where name like '%pt::%' order by start desc limit ?", (int(n),)).fetchall()[::-1]
    if not rows:
        return
    t0, last_end, busy, gaps = rows[0][1], None, 0, 0
    print(f"# last {len(rows)} dispatches")
    for name, st, en, gx, wx in rows:
        short = name.replace("void ", "").split("(")[0][-40:]
        gap = (st - last_end) / 1e3 if last_end is not None else 0.0
        print(f"{short:<40} start {(st - t0) / 1e3:>10.1f}  dur {(en - st) / 1e3:>9.1f}  gap {gap:>8.1f}  grid {gx // max(wx, 1)}")
        busy += en - st
        gaps += max(gap, 0.0)
        last_end = en if last_end is None else max(last_end, en)
    print(f"# span {(last_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, gaps {gaps:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], by_grid="--by-grid" in sys.argv[2:])
    for i, a in enumerate(sys.argv):
        if a == "--tail":
            tail(sys.argv[1], sys.argv[i + 1])
        if a == "--timeline":
            timeline(sys.argv[1], sys.argv[i + 1])
