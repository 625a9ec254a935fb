#!/bin/bash
# usage (GPU box, repo root): tools/pmc_persist.sh <tag>  -- SQ counters of k_persist against k_traverse4 (20-frame batches, one stream):
# separate rocprofv3 --pmc passes over tools/run_frames.py heightfield 40 20 1 [persist=1] -> gpurun_out/<tag>/summary_{persist,plain}.txt
TAG=${1:?tag}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { local mode=$1 name=$2; shift 2
  local extra=""; [ "$mode" = persist ] && extra="persist=1"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/$mode/$name -o $name -- python3 $ROOT/tools/run_frames.py heightfield 40 20 1 $extra > $OUT/$mode.$name.log 2>&1 || echo "pass $mode $name failed"; }
for mode in persist plain; do
  pass $mode sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
  pass $mode sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/persist k_persist > $OUT/summary_persist.txt 2>&1
python3 tools/pmc_summary.py $OUT/plain k_traverse4 > $OUT/summary_plain.txt 2>&1
for m in persist plain; do rm -rf $OUT/$m/*/*/*.db; done
cat $OUT/summary_persist.txt $OUT/summary_plain.txt
