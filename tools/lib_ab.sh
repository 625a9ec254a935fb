#!/bin/bash
# A/B of differently built libptcore.so files on one box: tools/lib_ab.sh "<tag> <waves...>" ...
set -e
cd cuda-path-tracer_amd
cp libptcore.so /tmp/libptcore_orig.so
for spec in "$@"; do
  set -- $spec
  tag=$1; shift
  cp libptcore_$tag.so libptcore.so
  (cd .. && timeout -k 10 200 python tools/occ_probe.py $tag "$@")
done
cp /tmp/libptcore_orig.so libptcore.so
