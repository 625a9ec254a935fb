#!/bin/bash
# round 4, first GPU pass: tests, the bench lines that gained parity legs, the baseline numbers VERDICT's targets refer to
TAG=${1:-r4a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.log 2>&1 || { tail -20 $OUT/bench_steps20.log; exit 1; }
timeout -k 10 300 python bench.py --config 2 > $OUT/bench_config2.log 2>&1 || { tail -20 $OUT/bench_config2.log; exit 1; }
timeout -k 10 300 python bench.py --config 5 > $OUT/bench_config5.log 2>&1 || { tail -20 $OUT/bench_config5.log; exit 1; }
timeout -k 10 400 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 20 --warmup 5 --no-extras > $OUT/bench_rehearse2.log 2>&1 || { tail -20 $OUT/bench_rehearse2.log; exit 1; }
timeout -k 10 300 python bench.py --share-of 8 --steps 20 --warmup 5 --no-extras > $OUT/bench_share8.log 2>&1 || { tail -20 $OUT/bench_share8.log; exit 1; }
for f in steps20 config2 config5 rehearse2 share8; do python3 - $OUT/bench_$f.log $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{"metric"')]
if not l: print(sys.argv[2], "no line"); sys.exit(0)
j = json.loads(l[-1]); r = j.get("roofline") or {}
print(sys.argv[2], j["value"], j["unit"], j["ms_per_step"], "frac", r.get("frac"), "fl", r.get("frame_level_frac"), "launch_us", r.get("avg_launch_us"), "parity", j.get("parity") and {k: v for k, v in j["parity"].items() if k in ("bit_exact","live_equal","rays_equal","denoised_max_abs_err","rgba_max_lsb")}, "cpu", (j.get("cpu_baseline") or {}).get("value"), "rccl", j.get("rccl_ranks"), j.get("distinct_devices"), "walked", j.get("mrays_per_s_walked"), "lat", j.get("latency"))
PY
done
