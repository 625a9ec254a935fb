#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-serial}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/prof -o s -- python3 $ROOT/tools/run_frames.py heightfield 16 1 1 > $OUT/run.log 2>&1
cd $ROOT
python3 tools/rocpd_summary.py "$(ls $OUT/prof/*.db | head -1)" > $OUT/kernel_stats.txt
python3 - "$(ls $OUT/prof/*.db | head -1)" <<'PY' > $OUT/timeline.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x from kernels order by start").fetchall()
# last full frame: find the last k_raygen
idx = [i for i, r in enumerate(rows) if 'k_raygen' in r[0]]
a = idx[-2]; b = idx[-1]
t0 = rows[a][1]
prev_end = t0
tot_k = 0; tot_gap = 0
for name, s, e, g in rows[a:b]:
    short = name.split('(')[0].replace('void pt::', '').replace('pt::', '')[:34]
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:8.1f}  grid {g:9d}  {short}")
    tot_k += e - s; tot_gap += max(0, s - prev_end); prev_end = e
print(f"frame: {(rows[b][1] - t0) / 1e3:.1f} us, kernels {tot_k / 1e3:.1f} us, gaps {tot_gap / 1e3:.1f} us")
PY
rm -rf $OUT/prof
cat $OUT/timeline.txt
