#!/bin/bash
# usage (GPU box, repo root): tools/profile_upload.sh <tag>
# rocprofv3 --kernel-trace --stats of tools/time_upload.py (scene upload of the benchmark mesh: GPU BVH build + layouts)
# -> gpurun_out/<tag>/{upload.log,kernel_stats.txt}
TAG=${1:?tag}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o upload -- python3 "$ROOT/tools/time_upload.py" > "$OUT/upload.log" 2>&1
cd "$ROOT"
python3 tools/rocpd_summary.py "$(ls "$OUT"/prof/*.db | head -1)" > "$OUT/kernel_stats.txt" 2>&1
rm -rf "$OUT/prof"
cat "$OUT/upload.log"; cat "$OUT/kernel_stats.txt"
