#!/bin/bash
# single-frame latency (bench.py's latency leg) against the sizing of small traversal launches
OUT=gpurun_out/${1:-r4lat}; mkdir -p $OUT
for p in "small_waves=3072" "small_waves=2048" "small_waves=2560" "small_waves=4096" "small_waves=2048 --param min_waves=512" "small_waves=3072 --param small_rays_per_lane=8" "small_waves=2048 --param small_rays_per_lane=8"; do
  tag=$(echo $p | tr -d ' =-' ); python3 bench.py --steps 20 --warmup 5 --cpu-frames 1 --param $p > $OUT/$tag.log 2>&1
  python3 - "$OUT/$tag.log" "$p" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(f'{sys.argv[2]:<55}', d["value"], d["latency"], d["parity"]["bit_exact"], d["steady_state"]["value"])
PY
done
