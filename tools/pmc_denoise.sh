#!/bin/bash
# usage (GPU box, repo root): tools/pmc_denoise.sh <tag> [calls=8]
# What binds k_denoise_lds<step> (review item 6 of round 4): separate rocprofv3 --pmc passes over tools/run_denoise.py
# (one traced frame, then <calls> four-pass denoise calls) -> gpurun_out/<tag>/pmc_sq.json + pmc_traffic.json for the
# kernel substring k_denoise_lds; copy pmc_sq.json to profiles/pmc_denoise.json (bench.py --config 5 reads it).
TAG=${1:?tag}; CALLS=${2:-8}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/$name -o $name -- python3 $ROOT/tools/run_denoise.py $CALLS > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
cd $ROOT
python3 tools/pmc_summary.py $OUT k_denoise_lds > $OUT/summary.txt 2>&1
python3 tools/pmc_json.py $OUT k_denoise_lds 1 >> $OUT/summary.txt 2>&1
for d in fetch write rdreq sq1 sq2 lds tcc tcp; do rm -rf $OUT/$d/*/*.db; done
cat $OUT/summary.txt
