#!/bin/bash
# usage (GPU box, repo root): tools/ab_timeline.sh <outdir> [bench args]   -- tools/profile_bench.sh (kernel table +
# timeline of the last 64 library dispatches) once per cuda-path-tracer_amd/libptcore_w_*.so and once with the default
# library: which kernel an experiment build moves (such builds need not be correct: bench.py's parity leg is off here)
OUT=${1:?outdir}; shift
for lib in cuda-path-tracer_amd/libptcore.so cuda-path-tracer_amd/libptcore_w_*.so; do
  [ -f "$lib" ] || continue
  tag=$(basename $lib .so)
  PTCORE_LIB=$PWD/$lib SUMMARY_FLAGS="--timeline 64" tools/profile_bench.sh $OUT/$tag "$@" > /dev/null 2>&1
  echo "== $tag"; grep "k_shade_fused\|k_raygen\|k_accumulate" gpurun_out/$OUT/$tag/kernel_stats.txt | grep "start" | tail -24 | head -12
done
