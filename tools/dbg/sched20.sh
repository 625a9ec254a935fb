#!/bin/bash
mkdir -p gpurun_out/r2n
for cfg in "2 10" "4 5" "1 20" "3 7" "5 4" "2 16"; do
  set -- $cfg
  for rep in 1 2; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --streams $1 --batch-frames $2 > gpurun_out/r2n/s$1b$2.log 2>&1
  echo "streams $1 batch $2: $(grep -o '"value": [0-9.]*' gpurun_out/r2n/s$1b$2.log | head -1) $(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r2n/s$1b$2.log | head -1)"
  done
done
