#!/bin/bash
# usage (GPU box, repo root): tools/r4_sched.sh <tag> -- the driver's 20 timed steps under other schedules (streams x frames per launch)
TAG=${1:?tag}; mkdir -p gpurun_out/$TAG
for rep in 1 2; do
for sch in "1 20" "2 10" "3 7" "4 5" "2 5"; do
  set -- $sch
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-frames 1 --no-extras --streams $1 --batch-frames $2 > gpurun_out/$TAG/s$1_b$2.log 2>&1
  python3 - gpurun_out/$TAG/s$1_b$2.log "$sch" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{"metric"')]
if not l: print(sys.argv[2], "FAILED"); sys.exit(0)
d = json.loads(l[-1]); p = d.get("parity") or {}
print(f'streams x frames {sys.argv[2]:<6} parity {p.get("bit_exact")} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:.4f} ms/step launch {d["roofline"]["avg_launch_us"]} us concurrent {d["roofline"]["concurrent_launches"]}')
PY
done; done
