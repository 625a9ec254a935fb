#!/bin/bash
# A/B of the default library against every libptcore_w_*.so except the tail-profile build: the three workloads the verdict's targets name
TAG=${1:-r4ab}; mkdir -p gpurun_out/$TAG
mv cuda-path-tracer_amd/libptcore_w_tail.so /tmp/ 2>/dev/null
echo "== driver command (--steps 20 --warmup 5)"; REPS=${REPS:-2} tools/ab.sh $TAG/s20 --steps 20 --warmup 5
echo "== share of 8"; REPS=${REPS:-2} tools/ab.sh $TAG/sh8 --share-of 8 --steps 20 --warmup 5
echo "== default run (256 steps, 2 x 32)"; REPS=1 tools/ab.sh $TAG/def
echo "== serial frame latency"
for lib in cuda-path-tracer_amd/libptcore.so cuda-path-tracer_amd/libptcore_w_*.so; do
  PTCORE_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/lat_$(basename $lib .so).log 2>&1
  python3 - gpurun_out/$TAG/lat_$(basename $lib .so).log $(basename $lib .so) <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{"metric"')]
j = json.loads(l[-1]); print(sys.argv[2], j["latency"], "steady", j["steady_state"]["value"], [ (b["bounce"], b["trace_ms"], b["frac"]) for b in j["roofline"]["per_bounce"]])
PY
done
mv /tmp/libptcore_w_tail.so cuda-path-tracer_amd/ 2>/dev/null
