#!/bin/bash
# one rank's eighth of the frame (--share-of 8, the driver's 20 steps) against wavefronts per launch and the schedule
OUT=gpurun_out/${1:-r4share8}; mkdir -p $OUT
for rep in 1 2; do
for cfg in "2 10 2560" "2 10 2048" "2 10 1536" "2 10 3072" "1 20 2560" "1 20 3584" "4 5 2048" "3 7 2560"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py --share-of 8 --steps 20 --warmup 5 --cpu-frames 1 --no-extras --streams $1 --batch-frames $2 --traverse-waves $3 > $OUT/s$1_b$2_w$3.log 2>&1
  python3 - $OUT/s$1_b$2_w$3.log "$cfg" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{"metric"')]
if not l: print(sys.argv[2], "FAILED"); sys.exit(0)
d = json.loads(l[-1]); p = d.get("parity") or {}
print(f'streams frames waves {sys.argv[2]:<12} parity {p.get("bit_exact")} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:.4f} ms/step')
PY
done; done
