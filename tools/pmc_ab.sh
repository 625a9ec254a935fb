#!/bin/bash
# usage (GPU box, repo root): tools/pmc_ab.sh <tag> <kernel-substring> [frames_per_launch=20] [frames=40]
# One --pmc pass (instruction and wave counters) over tools/run_frames.py per cuda-path-tracer_amd/libptcore*.so:
# what an experiment build changes in a kernel's instruction count, per dispatch (tools/pmc_bounce.py).
TAG=${1:?tag}; KERN=${2:?kernel}; FPL=${3:-20}; FRAMES=${4:-40}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for lib in $ROOT/cuda-path-tracer_amd/libptcore.so $ROOT/cuda-path-tracer_amd/libptcore_w_*.so; do
  [ -f "$lib" ] || continue
  name=$(basename $lib .so); OUT=$ROOT/gpurun_out/$TAG/$name; mkdir -p $OUT
  export PTCORE_LIB=$lib
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -f csv -d $OUT/sq1 -o sq1 -- python3 $ROOT/tools/run_frames.py heightfield $FRAMES $FPL 2 > $OUT/sq1.log 2>&1 || echo "pass failed"
  (cd $ROOT && python3 tools/pmc_bounce.py gpurun_out/$TAG/$name sq1 $KERN > $OUT/bounces.txt 2>&1)
  rm -rf $OUT/sq1/*/*.db
  echo "== $name"; head -9 $OUT/bounces.txt | cut -c1-260
done
