#!/bin/bash
# usage (GPU box, repo root): tools/r4_pmc_c2.sh <tag>   -- counter passes over config 2's scene; per-kernel means of its three bounce kernels
TAG=${1:?tag}
tools/pmc.sh $TAG 32 64 bunny > gpurun_out/$TAG.pmc.log 2>&1
for k in k_shade_fused k_spheres k_traverse4m; do python3 tools/pmc_summary.py gpurun_out/$TAG $k > gpurun_out/$TAG/summary_$k.txt 2>&1; done
python3 - <<PY
import json
for line in open("gpurun_out/$TAG/sq1.log"):
    if line.startswith("{"): print(line.strip())
PY
