#!/bin/bash
# usage (GPU box): tools/sweep.sh <outdir> <lib or ""> <param name> "<values>" [bench args...]  -- one bench.py run per value
OUT=gpurun_out/${1:?outdir}; LIB=$2; NAME=$3; VALUES=$4; shift 4
mkdir -p $OUT
for v in $VALUES; do
  PTCORE_LIB=${LIB:+$PWD/$LIB} timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --param $NAME=$v "$@" > $OUT/${NAME}_$v.log 2>&1
  python3 - "$OUT/${NAME}_$v.log" "$NAME=$v" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')]
if not line:
    print(sys.argv[2], "FAILED"); sys.exit(0)
d = json.loads(line[-1]); r = d["roofline"]
pb = " ".join(f'b{b["bounce"]}:{b["trace_ms"]:.2f}' for b in r["per_bounce"])
print(f'{sys.argv[2]:<22} {d["value"]:9.1f} Mrays/s  {d["ms_per_step"]:.4f} ms/step  trace_ms {pb}')
PY
done
