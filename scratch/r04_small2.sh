#!/bin/bash
# ping-pong passes of engines below the chain threshold: numbering x tile-local applications (first combination = warm-up, discard)
set -o pipefail
out=${1:-gpurun_out/r04f_small.txt}; : > $out
C="warmup=CWR_TILE_ORDER:hilbert lanes_r2=CWR_TILE_ORDER:lanes,CWR_LOCAL_REPS:2 hilbert_r2=CWR_TILE_ORDER:hilbert,CWR_LOCAL_REPS:2 hilbert_r3=CWR_TILE_ORDER:hilbert,CWR_LOCAL_REPS:3 hilbert_r4=CWR_TILE_ORDER:hilbert,CWR_LOCAL_REPS:4 hilbert_r6=CWR_TILE_ORDER:hilbert,CWR_LOCAL_REPS:6 hilbert_r8=CWR_TILE_ORDER:hilbert,CWR_LOCAL_REPS:8 default="
run() { timeout -k 10 300 python scratch/r04_small.py "$@" $C >> $out 2>&1 || echo "FAILED $* rc=$?" >> $out; }
run sq354 16; run sq245 16; run sq354 1; run sq245 4
MID_DT=400 run sq354 16; MID_DT=400 run sq354 1
run band200x50 12; run band160x50 12; run band160x50 1; run band109x28 12
tail -90 $out
