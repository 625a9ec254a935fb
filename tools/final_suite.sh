#!/bin/bash
# usage (GPU box, repo root): tools/final_suite.sh <tag>   -- the measurements DESIGN.md section 6 quotes, into gpurun_out/<tag>/:
# GPU tests, bench.py (driver's command, default run, configs 2 and 5, two-rank rehearsal), kernel tables under rocprofv3
TAG=${1:?tag}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -5 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.log 2>&1 || exit 1
timeout -k 10 300 python bench.py > $OUT/bench_default.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --config 2 > $OUT/bench_config2.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --config 5 > $OUT/bench_config5.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 20 --warmup 5 > $OUT/bench_rehearse2.log 2>&1 || exit 1
SUMMARY_FLAGS="--timeline 20" tools/profile_bench.sh $TAG/prof20 --steps 20 --warmup 5 > /dev/null 2>&1
tools/profile_bench.sh $TAG/profdef > /dev/null 2>&1
tools/profile_bench.sh $TAG/profc2 --config 2 > /dev/null 2>&1
for f in steps20 default config2 config5 rehearse2; do python3 - $OUT/bench_$f.log $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{"metric"')]
if not l: print(sys.argv[2], "no line"); sys.exit(0)
j = json.loads(l[-1]); r = j.get("roofline") or {}
print(sys.argv[2], j["value"], j["unit"], j["ms_per_step"], "frac", r.get("frac"), "launch_us", r.get("avg_launch_us"), "parity", (j.get("parity") or {}).get("bit_exact"))
PY
done
