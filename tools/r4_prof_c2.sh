#!/bin/bash
# kernel tables of config 2 (and config 3, 20 steps) under rocprofv3 for the default library and a variant
for lib in libptcore libptcore_w_noskip; do
  export PTCORE_LIB=$PWD/cuda-path-tracer_amd/$lib.so
  tools/profile_bench.sh r4prof/c2_$lib --config 2 --steps 64 --warmup 16 > /dev/null 2>&1
  tools/profile_bench.sh r4prof/s20_$lib --steps 20 --warmup 5 > /dev/null 2>&1
  echo "== $lib config 2"; head -12 gpurun_out/r4prof/c2_$lib/kernel_stats.txt | cut -c1-150
  echo "== $lib config 3 --steps 20"; head -10 gpurun_out/r4prof/s20_$lib/kernel_stats.txt | cut -c1-150
done
