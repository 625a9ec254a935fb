#!/bin/bash
# usage (GPU box, repo root): tools/r4_kinds.sh <tag> -- shade_kinds: frame tests, then default library against libptcore_w_*.so
TAG=${1:?tag}; mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_spheres.py tests/test_gpu_schedules.py tests/test_gpu_multimesh.py -x -q -m gpu > gpurun_out/$TAG/tests.log 2>&1 || { tail -30 gpurun_out/$TAG/tests.log; exit 1; }
tail -2 gpurun_out/$TAG/tests.log
REPS=2 tools/r4_ab3.sh $TAG
