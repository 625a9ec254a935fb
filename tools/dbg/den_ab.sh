#!/bin/bash
for lib in cuda-path-tracer_amd/libptcore.so cuda-path-tracer_amd/libptcore_w_*.so; do
  tag=$(basename $lib .so)
  PTCORE_LIB=$PWD/$lib python3 bench.py --config 5 --steps 48 --warmup 8 > gpurun_out/r2k/den_$tag.log 2>&1
  echo "$tag $(grep -o '"denoise_ms_per_pass": [0-9.]*\|"value": [0-9.]*' gpurun_out/r2k/den_$tag.log | tr '\n' ' ')"
done
