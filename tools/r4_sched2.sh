#!/bin/bash
# the driver's 20 steps as two 10-frame batches whose traversal launches share the chip (fewer wavefronts each)
OUT=gpurun_out/${1:-r4sched2}; mkdir -p $OUT
for rep in 1 2; do
for cfg in "1 20 5120" "2 10 2560" "2 10 3072" "2 10 3584" "2 10 2048"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-frames 1 --no-extras --streams $1 --batch-frames $2 --traverse-waves $3 > $OUT/s$1_b$2_w$3.log 2>&1
  python3 - $OUT/s$1_b$2_w$3.log "$cfg" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{"metric"')]
if not l: print(sys.argv[2], "FAILED"); sys.exit(0)
d = json.loads(l[-1]); p = d.get("parity") or {}
print(f'streams frames waves {sys.argv[2]:<12} parity {p.get("bit_exact")} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:.4f} ms/step')
PY
done; done
