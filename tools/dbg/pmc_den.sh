#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-pmcden}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/$name -o $name -- python3 $ROOT/tools/run_denoise.py 8 > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
pass sq3 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
cd $ROOT
python3 tools/pmc_summary.py $OUT k_denoise_lds > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
