#!/bin/bash
# round 4: resident passes (k_sq_resident, scratch/r04_resident.patch) against one launch per pass on engines below the chain threshold
set -o pipefail
out=gpurun_out/r04ah_resident.txt; : > $out
C="warmup= plain=CWR_NO_RESIDENT:1 resident= plain2=CWR_NO_RESIDENT:1 resident2="
run() { timeout -k 10 300 python scratch/r04_small.py "$@" $C >> $out 2>&1 || echo "FAILED $* rc=$?" >> $out; }
run sq245 16; run sq354 1; run sq245 4; run sq125 16
MID_DT=400 run sq245 16
run band200x50 12; run band160x50 1
grep -v "^\[cwr\]" $out | tail -60
