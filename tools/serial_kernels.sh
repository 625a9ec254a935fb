#!/bin/bash
# usage (GPU box, repo root): tools/serial_kernels.sh <tag> <scene> <frames per launch> <frames> [name=value ...]
# One rocprofv3 --pmc pass (the profiler serialises the kernels: every duration is the kernel's own work, nothing waits for
# wavefront slots beside another stream's launch) -> per-kernel table of mean durations, gpurun_out/<tag>/serial_kernels.txt
TAG=${1:?tag}; SCENE=${2:-bunny}; FPL=${3:-32}; FRAMES=${4:-64}; shift 4
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES -f csv -d $OUT/ser -o ser -- python3 $ROOT/tools/run_frames.py $SCENE $FRAMES $FPL 2 "$@" > $OUT/ser.log 2>&1 || echo "pass failed"
cd $ROOT
python3 - $OUT/ser/ser_counter_collection.csv "$SCENE $FRAMES frames, $FPL per launch, $*" > $OUT/serial_kernels.txt <<'PY'
import csv, sys, collections
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    dur[name][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"# kernels serialised by the profiler (one --pmc pass): {sys.argv[2]}")
print(f"{'kernel':<60} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'max_us':>10}")
for name, d in sorted(dur.items(), key=lambda kv: -sum(kv[1].values())):
    if name.startswith("pt::k_lay") or name.startswith("pt::k_bins") or "rocclr" in name: continue
    v = list(d.values())
    print(f"{name[-60:]:<60} {len(v):>6} {sum(v)/1e6:>10.3f} {sum(v)/len(v)/1e3:>10.2f} {max(v)/1e3:>10.2f}")
PY
rm -rf $OUT/ser/*.db
cat $OUT/serial_kernels.txt
