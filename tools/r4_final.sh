#!/bin/bash
# round 4, final measurements: counters (two launch shapes), the bench lines, kernel tables under rocprofv3
TAG=${1:-r4final}; OUT=gpurun_out/$TAG; mkdir -p $OUT
tools/pmc.sh $TAG/pmc20 20 40 > $OUT/pmc20.log 2>&1
tools/pmc.sh $TAG/pmc32 32 64 > $OUT/pmc32.log 2>&1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.log 2>&1 || { tail -20 $OUT/bench_steps20.log; exit 1; }
timeout -k 10 300 python bench.py > $OUT/bench_default.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --config 2 > $OUT/bench_config2.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --config 5 > $OUT/bench_config5.log 2>&1 || exit 1
timeout -k 10 400 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 20 --warmup 5 --no-extras > $OUT/bench_rehearse2.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --share-of 8 --steps 20 --warmup 5 --no-extras > $OUT/bench_share8.log 2>&1 || exit 1
SUMMARY_FLAGS="--timeline 20" tools/profile_bench.sh $TAG/prof20 --steps 20 --warmup 5 > /dev/null 2>&1
tools/profile_bench.sh $TAG/profdef > /dev/null 2>&1
tools/profile_bench.sh $TAG/profc2 --config 2 > /dev/null 2>&1
tools/profile_bench.sh $TAG/profc5 --config 5 > /dev/null 2>&1
for f in steps20 default config2 config5 rehearse2 share8; do python3 - $OUT/bench_$f.log $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{"metric"')]
if not l: print(sys.argv[2], "no line"); sys.exit(0)
j = json.loads(l[-1]); r = j.get("roofline") or {}
print(sys.argv[2], j["value"], j["unit"], j["ms_per_step"], "frac", r.get("frac"), "fl", r.get("frame_level_frac"), "launch_us", r.get("avg_launch_us"), "parity", j.get("parity") and {k: v for k, v in j["parity"].items() if k in ("bit_exact","live_equal","rays_equal","denoised_max_abs_err","rgba_max_lsb")}, "cpu", (j.get("cpu_baseline") or {}).get("value"), "walked", j.get("mrays_per_s_walked"), "lat", j.get("latency"), "steady", (j.get("steady_state") or {}).get("value"))
PY
done
