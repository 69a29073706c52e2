#!/bin/bash
set -o pipefail
out=gpurun_out/r04ao_lane_mix.txt; : > $out
C="warmup= straight=CWR_LANE_KIND:straight channel=CWR_LANE_KIND:channel q_x=CWR_LANE_MIX:q_x y_sigma=CWR_LANE_MIX:y_sigma y_x_shift=CWR_LANE_MIX:y_x_shift straight2=CWR_LANE_KIND:straight"
run() { timeout -k 10 400 python scratch/r04_small.py "$@" $C >> $out 2>&1 || echo "FAILED $* rc=$?" >> $out; }
run bend1026x256@0 16
run bend1026x512@0 16
grep -v "^\[cwr\]\|Warn" $out
