#!/bin/bash
# tools/lib_tuner.sh <passes> "<libtag> <tuner args...>" ...  -- tuner.py under differently built libraries, in rotation
passes=$1; shift
cp cuda-path-tracer_amd/libptcore.so /tmp/libptcore_orig.so
for p in $(seq $passes); do
  for spec in "$@"; do
    read -r tag rest <<< "$spec"
    cp cuda-path-tracer_amd/libptcore_$tag.so cuda-path-tracer_amd/libptcore.so
    echo "== $tag"
    timeout -k 10 300 python tools/tuner.py 1 $rest | grep -v distinct
  done
done
cp /tmp/libptcore_orig.so cuda-path-tracer_amd/libptcore.so
