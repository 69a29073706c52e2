#!/bin/bash
# same-box sweep: engines below the chain threshold, rounds per launch (round 4, task 1b).  usage: r04_small.sh <out> [cases...]
set -o pipefail
out=${1:-gpurun_out/r04e_small.txt}; : > $out
C="base=CWR_VISITS:1 base_hilbert=CWR_VISITS:1,CWR_TILE_ORDER:hilbert v2=CWR_VISITS:2 v4=CWR_VISITS:4 v8=CWR_VISITS:8 v16=CWR_VISITS:16 v32=CWR_VISITS:32 v8_r4=CWR_VISITS:8,CWR_LOCAL_REPS:4 v8_r3=CWR_VISITS:8,CWR_LOCAL_REPS:3 v8_hilbert=CWR_VISITS:8,CWR_TILE_ORDER:hilbert v8_b2=CWR_VISITS:8,CWR_TCL_BLOCKS_PER_CU:2"
for cs in "sq354 16" "sq245 16" "sq354 1" "band200x50 12" "band160x50 12" "band160x50 1"; do
  timeout -k 10 300 python scratch/r04_small.py $cs $C >> $out 2>&1 || echo "FAILED $cs rc=$?" >> $out
done
tail -70 $out
