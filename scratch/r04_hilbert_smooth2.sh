#!/bin/bash
set -o pipefail
out=gpurun_out/r04aw_hilbert_smooth2.txt; : > $out
C="warmup= s0=CWR_HILBERT_SMOOTH:0 s16=CWR_HILBERT_SMOOTH:16 s32=CWR_HILBERT_SMOOTH:32 s64=CWR_HILBERT_SMOOTH:64 s128=CWR_HILBERT_SMOOTH:128"
run() { timeout -k 10 400 python scratch/r04_small.py "$@" $C >> $out 2>&1 || echo "FAILED $* rc=$?" >> $out; }
run band200x50 12; run band160x50 1; run band160x50 12; run sq354 1; run sq125 16
grep -v "^\[cwr\]\|Warn\|warmup" $out
