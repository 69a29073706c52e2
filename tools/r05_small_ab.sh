#!/bin/bash
# round 5: what the step-opening fusion and the copy-free check buy an engine of one rank's size (119 k cells), and a sweep of the
# lane length / tile-local applications there.  Usage (GPU box): bash tools/r05_small_ab.sh <tag>
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05f}
out=gpurun_out/${tag}_small_ab.txt; : > $out
run() { timeout -k 10 500 python tools/r04_small.py "$@" >> $out 2>&1 || echo "FAILED $* rc=$?" >> $out; }
AB="warmup= r04=CWR_NO_NOTE:1,CWR_NO_FUSED_BEGIN:1 note_only=CWR_NO_FUSED_BEGIN:1 fused_only=CWR_NO_NOTE:1 r05= r04b=CWR_NO_NOTE:1,CWR_NO_FUSED_BEGIN:1 r05b="
run sq354 16 $AB
run sq354 1 $AB
run sq245 16 $AB
run sq1000 16 warmup= r04=CWR_NO_NOTE:1,CWR_NO_FUSED_BEGIN:1 r05= r04b=CWR_NO_NOTE:1,CWR_NO_FUSED_BEGIN:1 r05b=
SW="warmup= len3x2= len4x2=CWR_LANE_LEN:4 len6x2=CWR_LANE_LEN:6 len3x3=CWR_LOCAL_REPS:3 len4x3=CWR_LANE_LEN:4,CWR_LOCAL_REPS:3 len6x3=CWR_LANE_LEN:6,CWR_LOCAL_REPS:3 len6x4=CWR_LANE_LEN:6,CWR_LOCAL_REPS:4 len8x4=CWR_LANE_LEN:8,CWR_LOCAL_REPS:4"
run sq354 16 $SW
grep -v "^\[cwr\]\|Warn\|warmup" $out
