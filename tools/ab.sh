#!/bin/bash
# tools/ab.sh "<EXTRA params or ->  <F,B,waves>..." ...   -- occ_probe runs with the current library
for spec in "$@"; do
  set -- $spec
  extra=$1; shift
  [ "$extra" = "-" ] && extra=""
  EXTRA=$extra timeout -k 10 200 python tools/occ_probe.py "[$extra]" "$@" || exit 1
done
