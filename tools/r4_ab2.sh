#!/bin/bash
# default library against libptcore_w_old.so (and any other variant), the three workloads, REPS times
TAG=${1:-r4ab}; mkdir -p gpurun_out/$TAG
echo "== driver command (--steps 20 --warmup 5)"; REPS=${REPS:-2} tools/ab.sh $TAG/s20 --steps 20 --warmup 5
echo "== share of 8"; REPS=${REPS:-2} tools/ab.sh $TAG/sh8 --share-of 8 --steps 20 --warmup 5
echo "== default run (256 steps, 2 x 32)"; REPS=${REPS:-2} tools/ab.sh $TAG/def
