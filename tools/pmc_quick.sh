#!/bin/bash
# usage (GPU box, repo root): tools/pmc_quick.sh <tag> [frames_per_launch=20] [frames=40] [name=value ...]
# Two --pmc passes only (fabric read requests by size, L2 hits) and the per-dispatch view of tools/pmc_bounce.py:
# which bounce's traversal launch the traffic belongs to.
TAG=${1:?tag}; FPL=${2:-20}; FRAMES=${3:-40}; shift 3
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/$name -o $name -- python3 $ROOT/tools/run_frames.py heightfield $FRAMES $FPL 2 $EXTRA > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
EXTRA="$*"
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
cd $ROOT
python3 tools/pmc_bounce.py $OUT rdreq k_traverse4 > $OUT/bounces.txt 2>&1
python3 tools/pmc_bounce.py $OUT tcc k_traverse4 >> $OUT/bounces.txt 2>&1
for d in rdreq tcc; do rm -rf $OUT/$d/*/*.db; done
head -12 $OUT/bounces.txt
