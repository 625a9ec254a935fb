#!/bin/bash
mkdir -p gpurun_out/r2q
for w in 5120 4096 4608 3584 6144; do for st in 2 3; do
  python3 bench.py --no-cpu-baseline --no-extras --traverse-waves $w --streams $st > gpurun_out/r2q/w$w.log 2>&1
  echo "waves $w streams $st: $(grep -o '"value": [0-9.]*' gpurun_out/r2q/w$w.log | head -1) $(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r2q/w$w.log | head -1)"
done; done
