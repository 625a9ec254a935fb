#!/bin/bash
# usage (on the GPU box): tools/ab.sh <outdir> <bench args...>   -- runs bench.py once per cuda-path-tracer_amd/libptcore_w_*.so
# (built by `make -C cuda-path-tracer_amd/csrc variant TAG=.. EXTRA=..`) and once with the default library
OUT=gpurun_out/${1:?outdir}; shift
mkdir -p $OUT
REPS=${REPS:-1}
for rep in $(seq $REPS); do
for lib in cuda-path-tracer_amd/libptcore.so cuda-path-tracer_amd/libptcore_w_*.so; do
  [ -f "$lib" ] || continue
  tag=$(basename $lib .so)
  # the parity leg stays on (one oracle frame, ~2 s): a variant that corrupts its results must not produce a number
  PTCORE_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --cpu-frames 1 --no-extras "$@" > $OUT/$tag.log 2>&1
  python3 - "$OUT/$tag.log" "$tag" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')]
if not line:
    print(sys.argv[2], "FAILED"); sys.exit(0)
d = json.loads(line[-1]); r = d["roofline"]
par = d.get("parity")
if par is not None and not (par["bit_exact"] and par["live_equal"] and par["rays_equal"]):
    print(sys.argv[2], "PARITY FAILED -- number void:", par); sys.exit(0)
print(f'{sys.argv[2]:<28} {"parity ok" if par else "parity n/a"} {d["value"]:9.1f} Mrays/s  {d["ms_per_step"]:.4f} ms/step  launch {r["avg_launch_us"]:8.1f} us  frac {r["frac"]:.4f}  nodes/ray {r["node_visits_per_ray"]}  box/ray {r["box_tests_per_ray"]}  tri/ray {r["tri_tests_per_ray"]}')
PY
done
done
