#!/bin/bash
# "beam" on / off (runtime parameter): the three workloads + serial frame latency
TAG=${1:-r4beam}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { # name, args...
  local name=$1; shift
  timeout -k 10 300 python3 bench.py --cpu-frames 1 "$@" > $OUT/$name.log 2>&1
  python3 - $OUT/$name.log $name <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{"metric"')]
if not l: print(sys.argv[2], "FAILED"); print(open(sys.argv[1]).read()[-800:]); sys.exit(0)
d = json.loads(l[-1]); r = d["roofline"]; p = d.get("parity") or {}
ok = p.get("bit_exact") and p.get("live_equal") and p.get("rays_equal")
print(f'{sys.argv[2]:<14} {"parity ok" if ok else "PARITY FAILED"} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:.4f} ms/step launch {r["avg_launch_us"]:8.1f} us frac {r["frac"]:.4f} nodes/ray {r["node_visits_per_ray"]}', [(b["bounce"], b["trace_ms"], b["node_visits_per_ray"]) for b in r["per_bounce"][:3]], d.get("latency"))
PY
}
for rep in 1 2; do
run s20_on --steps 20 --warmup 5 --no-extras
run s20_off --steps 20 --warmup 5 --no-extras --param beam=0
run sh8_on --share-of 8 --steps 20 --warmup 5 --no-extras
run sh8_off --share-of 8 --steps 20 --warmup 5 --no-extras --param beam=0
done
run def_on --no-extras
run def_off --no-extras --param beam=0
run lat_on --steps 20 --warmup 5
