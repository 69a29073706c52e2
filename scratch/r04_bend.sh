#!/bin/bash
# round 4: lanes that follow a meander (ordering.channel_coordinates) against straight lanes and the Hilbert curve, chained passes
set -o pipefail
out=gpurun_out/r04ak_bend.txt; : > $out
C="warmup= straight=CWR_LANE_KIND:straight channel=CWR_LANE_KIND:channel auto= hilbert=CWR_TILE_ORDER:hilbert"
run() { timeout -k 10 400 python scratch/r04_small.py "$@" $C >> $out 2>&1 || echo "FAILED $* rc=$?" >> $out; }
run bend1026x256@0 16
run bend1026x256@1.0 16
BEND_WAVELENGTH=5130 run bend1026x256@0.55 16
run bend1026x256@1.0 1
MID_DT=400 run bend1026x256@1.0 16
grep -v "^\[cwr\]\|Warn" $out
