#!/bin/bash
TAG=$1; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
pass() { local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/$name -o $name -- python3 $ROOT/tools/run_frames.py bunny 64 32 2 > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cd $ROOT
python3 tools/pmc_summary.py $OUT "k_shade_fused<true, false, true>" > $OUT/summary_shade.txt 2>&1
for d in sq1 sq2 fetch write; do rm -rf $OUT/$d/*/*.db; done
cat $OUT/summary_shade.txt
