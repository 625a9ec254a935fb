#!/bin/bash
# refill_lanes / static_eighths: (32, 4) against (20, 3) on every bench configuration (ptc_set_param, one build)
OUT=gpurun_out/${1:-r4feed}; mkdir -p $OUT
run() { local tag=$1; shift
  python3 bench.py --cpu-frames 1 "$@" > $OUT/$tag.log 2>&1
  python3 - $OUT/$tag.log "$tag" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric'):
        d=json.loads(l); print(f'{sys.argv[2]:<28}', d["value"], d["unit"], d["ms_per_step"], (d.get("parity") or {}).get("bit_exact"), d.get("latency"), (d.get("steady_state") or {}).get("value"))
PY
}
for rep in 1 2; do
for p in "32 4" "20 3" "28 4" "32 3"; do set -- $p; P="--param refill_lanes=$1 --param static_eighths=$2"
  run s20_$1_$2 --steps 20 --warmup 5 $P
  run c2_$1_$2 --config 2 $P
  run share8_$1_$2 --share-of 8 --steps 20 --warmup 5 --no-extras $P
  run c5_$1_$2 --config 5 $P
done; done
