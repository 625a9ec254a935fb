#!/bin/bash
# default library against the variants: config 3 (driver's command, default run) and config 2
TAG=${1:-r4ab}; mkdir -p gpurun_out/$TAG
echo "== driver command (--steps 20 --warmup 5)"; REPS=${REPS:-2} tools/ab.sh $TAG/s20 --steps 20 --warmup 5
echo "== default run (256 steps, 2 x 32)"; REPS=${REPS:-2} tools/ab.sh $TAG/def
echo "== config 2"; for rep in $(seq ${REPS:-2}); do for lib in cuda-path-tracer_amd/libptcore.so cuda-path-tracer_amd/libptcore_w_*.so; do
  PTCORE_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --config 2 --cpu-frames 1 > gpurun_out/$TAG/c2_$(basename $lib .so).log 2>&1
  python3 - gpurun_out/$TAG/c2_$(basename $lib .so).log $(basename $lib .so) <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{"metric"')]
if not l: print(sys.argv[2], "FAILED"); sys.exit(0)
d = json.loads(l[-1]); p = d.get("parity") or {}
print(f'{sys.argv[2]:<24} parity {p.get("bit_exact")} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:.4f} ms/step  frame-level frac {d["roofline"].get("frame_level_frac")}')
PY
done; done
