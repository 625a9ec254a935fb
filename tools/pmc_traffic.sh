#!/bin/bash
# HBM-side traffic of the trace kernel: separate --pmc passes (kernel-trace only), see MI355X_MICROARCH.md "HBM".
OUT=/root/repo/gpurun_out/${1:-pmc_traffic}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { local name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/$name -o $name -- python3 /root/repo/tools/run_frames.py heightfield 3 64 > $OUT/$name.log 2>&1; }
pass fetch FETCH_SIZE && pass write WRITE_SIZE && pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum && pass wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum
ls $OUT
