#!/bin/bash
# usage (GPU box, repo root): tools/pmc_mem.sh <tag> [frames_per_launch=32] [frames=64] [kernel=k_traverse4]
# Texture-addresser / L1 / TLB counters of the traversal kernel (separate rocprofv3 --pmc passes, kernel-trace only):
# is the loop bound by the per-line rate of the L1 (each lane fetches its own 64-byte node), by L2 latency, or by
# address translation?  -> gpurun_out/<tag>/summary_mem.txt
TAG=${1:?tag}; FPL=${2:-32}; FRAMES=${3:-64}; KERNEL=${4:-k_traverse4}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/$name -o $name -- python3 $ROOT/tools/run_frames.py heightfield $FRAMES $FPL > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
pass ta1 TA_BUSY_avr TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
pass tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum
pass tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass tcp3 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass tlb1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
pass tlb2 TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum
cd $ROOT
python3 tools/pmc_summary.py $OUT $KERNEL > $OUT/summary_mem.txt 2>&1
for d in ta1 ta2 tcp1 tcp2 tcp3 tlb1 tlb2; do rm -rf $OUT/$d/*/*.db; done
cat $OUT/summary_mem.txt
