#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag> [bench.py arguments...]
# rocprofv3 --kernel-trace --stats of one bench.py run -> gpurun_out/<tag>/{bench.log,kernel_stats.txt}
# (rocprofv3 gets the program itself after `--`: no env / bash -c hop, see the GPU-box rules)
TAG=${1:?tag}; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras "$@" > "$OUT/bench.log" 2>&1
rc=$?
cd "$ROOT"
python3 tools/rocpd_summary.py "$(ls "$OUT"/prof/*.db | head -1)" $SUMMARY_FLAGS --tail "k_traverse4<false:8" > "$OUT/kernel_stats.txt" 2>&1
rm -rf "$OUT/prof"
tail -c 2500 "$OUT/bench.log"; echo; cat "$OUT/kernel_stats.txt"
exit $rc
