#!/bin/bash
# usage: tools/pmc_sq.sh <tag> <scene> <variant>   -- SQ counters only (two passes)
TAG=${1:-pmc}; SC=${2:-heightfield}; VAR=${3:-3}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU -f csv -d $OUT/sq1 -o sq1 -- python3 /root/repo/tools/run_frames.py $SC $VAR 16 > $OUT/sq1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY -f csv -d $OUT/sq2 -o sq2 -- python3 /root/repo/tools/run_frames.py $SC $VAR 16 > $OUT/sq2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum -f csv -d $OUT/tcp -o tcp -- python3 /root/repo/tools/run_frames.py $SC $VAR 16 > $OUT/tcp.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -f csv -d $OUT/tcc -o tcc -- python3 /root/repo/tools/run_frames.py $SC $VAR 16 > $OUT/tcc.log 2>&1
ls $OUT
