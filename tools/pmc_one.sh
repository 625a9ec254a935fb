#!/bin/bash
# usage: tools/pmc_one.sh <tag> <scene> <variant> <counters...>   (one --pmc pass)
TAG=$1; SC=$2; VAR=$3; shift 3
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/p -o p -- python3 /root/repo/tools/run_frames.py $SC $VAR 8 > $OUT/p.log 2>&1
ls $OUT/p | head -3
