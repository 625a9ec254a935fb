#!/bin/bash
# usage (GPU box, repo root): tools/r4_fold2.sh <tag> -- sphere_fold in the kernel that ends a bounce: the default library (fold
# compiled into k_shade_fused) against libptcore_w_nofoldtail.so (not compiled in), and sphere_lanes 0 (the fold takes the
# trailing run) against 1 (per-lane candidates) on the default library
TAG=${1:?tag}; mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -m pytest tests/test_gpu_spheres.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/$TAG/tests.log 2>&1 || { tail -30 gpurun_out/$TAG/tests.log; exit 1; }
tail -2 gpurun_out/$TAG/tests.log
REPS=2 tools/r4_ab3.sh $TAG
echo "== sphere_lanes 0 on the default library"
for rep in 1 2; do
  REPS=1 tools/ab.sh $TAG/s20_lanes0 --steps 20 --warmup 5 --param sphere_lanes=0 | grep "^libptcore "
  python3 bench.py --config 2 --cpu-frames 1 --param sphere_lanes=0 > gpurun_out/$TAG/c2_lanes0.log 2>&1
  python3 - <<PY
import json
for line in open("gpurun_out/$TAG/c2_lanes0.log"):
    if line.startswith('{"metric'):
        d=json.loads(line); print("config2 sphere_lanes=0", d["value"], d["ms_per_step"], d["parity"]["bit_exact"], d["roofline"]["frame_level_frac"])
PY
done
