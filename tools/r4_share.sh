#!/bin/bash
# share_rays on / off (runtime parameter), variants, and the pre-change library, on the workloads of the verdict's targets
TAG=${1:-r4share}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { # name, lib, args...
  local name=$1 lib=$2; shift 2
  PTCORE_LIB=$PWD/cuda-path-tracer_amd/$lib timeout -k 10 200 python3 bench.py --cpu-frames 1 --no-extras "$@" > $OUT/$name.log 2>&1
  python3 - $OUT/$name.log $name <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{"metric"')]
if not l: print(sys.argv[2], "FAILED"); print(open(sys.argv[1]).read()[-800:]); sys.exit(0)
d = json.loads(l[-1]); r = d["roofline"]; p = d.get("parity") or {}
ok = p.get("bit_exact") and p.get("live_equal") and p.get("rays_equal")
print(f'{sys.argv[2]:<24} {"parity ok" if ok else "PARITY FAILED"} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:.4f} ms/step launch {r["avg_launch_us"]:8.1f} us frac {r["frac"]:.4f}', [(b["bounce"], b["trace_ms"]) for b in r["per_bounce"]])
PY
}
for rep in 1 2; do
run s20_on libptcore.so --steps 20 --warmup 5
run s20_a4 libptcore_w_a4.so --steps 20 --warmup 5
run s20_off libptcore.so --steps 20 --warmup 5 --param share_rays=0
run s20_old libptcore_w_old.so --steps 20 --warmup 5
run sh8_on libptcore.so --share-of 8 --steps 20 --warmup 5
run sh8_a4 libptcore_w_a4.so --share-of 8 --steps 20 --warmup 5
run sh8_off libptcore.so --share-of 8 --steps 20 --warmup 5 --param share_rays=0
run sh8_old libptcore_w_old.so --share-of 8 --steps 20 --warmup 5
done
run def_on libptcore.so
run def_off libptcore.so --param share_rays=0
run def_old libptcore_w_old.so
