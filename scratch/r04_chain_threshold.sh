#!/bin/bash
# round 4, after the smoothed lane boundaries: from how many tiles per block do lanes + chains beat the Hilbert curve + ping-pong passes?
set -o pipefail
out=gpurun_out/r04ay_chain_threshold.txt; : > $out
C="warmup= hilbert=CWR_TILE_ORDER:hilbert lanes_min1=CWR_TILE_ORDER:lanes,CWR_CHAIN_MIN_TILES:1 hilbert2=CWR_TILE_ORDER:hilbert lanes_min1b=CWR_TILE_ORDER:lanes,CWR_CHAIN_MIN_TILES:1"
run() { timeout -k 10 400 python scratch/r04_small.py "$@" $C >> $out 2>&1 || echo "FAILED $* rc=$?" >> $out; }
run sq270 16; run sq290 16; run sq320 16; run sq400 16
run sq400 4; run sq500 4; run sq600 1; run sq700 1
run sq320 12
grep -v "^\[cwr\]\|Warn\|warmup" $out
