#!/bin/bash
set -o pipefail
out=gpurun_out/r04av_hilbert_smooth.txt; : > $out
C="warmup= s0=CWR_HILBERT_SMOOTH:0 s4=CWR_HILBERT_SMOOTH:4 s16=CWR_HILBERT_SMOOTH:16 s0b=CWR_HILBERT_SMOOTH:0 s16b=CWR_HILBERT_SMOOTH:16"
run() { timeout -k 10 400 python scratch/r04_small.py "$@" $C >> $out 2>&1 || echo "FAILED $* rc=$?" >> $out; }
run sq354 16; run sq245 16; run sq354 1; run band200x50 12; run band160x50 1
MID_DT=400 run sq354 16
grep -v "^\[cwr\]\|Warn\|warmup" $out
