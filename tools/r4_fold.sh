#!/bin/bash
# usage (GPU box, repo root): tools/r4_fold.sh <tag>  -- sphere_fold: sphere tests, then config 2 and config 3 with the parameter on / off
TAG=${1:?tag}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_spheres.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
for rep in 1 2; do
for v in 1 0; do
  timeout -k 10 300 python bench.py --config 2 --cpu-frames 1 --param sphere_fold=$v > $OUT/c2_fold${v}_$rep.log 2>&1 || { tail -5 $OUT/c2_fold${v}_$rep.log; exit 1; }
  python - <<PY
import json
for line in open("$OUT/c2_fold${v}_$rep.log"):
    if line.startswith('{"metric'):
        d=json.loads(line); print("config2 sphere_fold=$v", d["value"], d["ms_per_step"], d["parity"]["bit_exact"], d["roofline"]["frame_level_frac"])
PY
done; done
