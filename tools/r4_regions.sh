#!/bin/bash
# interleaved regions of the ray feed: schedule tests, tail profile, then default library against libptcore_w_contig.so
TAG=${1:?tag}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_schedules.py tests/test_gpu_multimesh.py tests/test_gpu_interleaved.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
PTCORE_LIB=$PWD/cuda-path-tracer_amd/libptcore_w_tail.so timeout -k 10 200 python3 tools/tailprof.py --frames 20 > $OUT/tail20.txt 2>&1 || { tail $OUT/tail20.txt; exit 1; }
grep "^    [0-7] " $OUT/tail20.txt | cut -c1-130
mv cuda-path-tracer_amd/libptcore_w_tail.so /tmp/
REPS=2 tools/r4_ab3.sh $TAG
