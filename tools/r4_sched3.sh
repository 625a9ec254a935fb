#!/bin/bash
# the driver's 20 steps as one large batch and one small one on a second stream (the small one in the large one's idle ends)
OUT=gpurun_out/${1:-r4sched3}; mkdir -p $OUT
for rep in 1 2; do
for cfg in "1 20" "2 18" "2 17" "2 16" "2 15" "2 14" "2 12"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-frames 1 --no-extras --streams $1 --batch-frames $2 > $OUT/s$1_b$2.log 2>&1
  python3 - $OUT/s$1_b$2.log "$cfg" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{"metric"')]
if not l: print(sys.argv[2], "FAILED"); sys.exit(0)
d = json.loads(l[-1]); p = d.get("parity") or {}
print(f'streams x frames per batch {sys.argv[2]:<6} parity {p.get("bit_exact")} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:.4f} ms/step')
PY
done; done
