#!/bin/bash
TAG=${1:-r4final3}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.log 2>&1 || exit 1
timeout -k 10 300 python bench.py > $OUT/bench_default.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o bench -- python3 $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OLDPWD/$OUT/bench_steps20_under_rocprof.log 2>&1
cd $OLDPWD
python3 tools/rocpd_summary.py "$(ls $OUT/prof/*.db | head -1)" --timeline 20 --tail "k_traverse4<false:8" > $OUT/kernel_stats_steps20.txt 2>&1
rm -rf $OUT/prof
grep -h '^{"metric"' $OUT/bench_steps20.log $OUT/bench_default.log $OUT/bench_steps20_under_rocprof.log | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); r = j['roofline']
    print(j['value'], j['ms_per_step'], 'frac', r['frac'], 'launch', r['avg_launch_us'], 'traffic', r['traffic'], 'valu', (r.get('valu') or {}).get('valu_pipe_frac'), 'parity', (j.get('parity') or {}).get('bit_exact'), 'steady', (j.get('steady_state') or {}).get('value'))
"
tail -3 $OUT/kernel_stats_steps20.txt
