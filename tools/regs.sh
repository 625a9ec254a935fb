#!/bin/bash
# Register / spill / occupancy report of the kernels of one translation unit, compiled with the library's own flags:
#   tools/regs.sh pt_shade.hip [filter] [extra -D flags]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/cuda-path-tracer_amd/csrc/$1; FILTER=${2:-.}; shift; shift || true
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-slp-vectorize "$@" -c "$SRC" -o $TMP/x.o -Rpass-analysis=kernel-resource-usage 2> $TMP/r.txt || { cat $TMP/r.txt; exit 1; }
grep -A12 "Function Name" $TMP/r.txt | grep -E "Function Name| VGPRs:|AGPRs:|Occupancy|Spill|ScratchSize|LDS Size" | sed 's/.*remark: //; s/ *\[-Rpass.*//' |
  awk '/Function Name/{printf "\n%s ", $3} !/Function Name/{printf "| %s ", $0}' | while read -r name rest; do
    [ -z "$name" ] && continue; echo "$(echo "$name" | c++filt | cut -c1-70) $rest"; done | grep -E "$FILTER"
rm -rf $TMP
