#!/bin/bash
OUT=gpurun_out/${1:-r4tail}; mkdir -p $OUT
export PTCORE_LIB=$PWD/cuda-path-tracer_amd/libptcore_w_tail.so
timeout -k 10 200 python3 tools/tailprof.py --frames 1 --fif 1 > $OUT/serial.txt 2>&1 || { tail $OUT/serial.txt; exit 1; }
timeout -k 10 200 python3 tools/tailprof.py --frames 20 > $OUT/steps20.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/tailprof.py --frames 10 --share-of 8 > $OUT/share8.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/tailprof.py --frames 1 --fif 1 --param split_idle=0 > $OUT/serial_nosplit.txt 2>&1 || exit 1
cat $OUT/serial.txt $OUT/steps20.txt $OUT/share8.txt $OUT/serial_nosplit.txt
