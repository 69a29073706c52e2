#!/bin/bash
# round 4, after the smoothed lane boundaries: engines below the chain threshold once more -- lanes + chains over short lists, smaller grids
set -o pipefail
out=gpurun_out/r04ax_small_lanes_again.txt; : > $out
C="warmup= hilbert= lanes_min1=CWR_TILE_ORDER:lanes,CWR_CHAIN_MIN_TILES:1 lanes_min1_det=CWR_TILE_ORDER:lanes,CWR_CHAIN_MIN_TILES:1,CWR_DET_DEFAULT_K:99 lanes_g512=CWR_TILE_ORDER:lanes,CWR_CHAIN_MIN_TILES:1,CWR_TCL_GRID:512 lanes_g768=CWR_TILE_ORDER:lanes,CWR_CHAIN_MIN_TILES:1,CWR_TCL_GRID:768 lanes_pp=CWR_TILE_ORDER:lanes,CWR_NO_CHAINS:1 hilbert2="
run() { timeout -k 10 400 python scratch/r04_small.py "$@" $C >> $out 2>&1 || echo "FAILED $* rc=$?" >> $out; }
run sq354 16; run sq245 16; run sq354 1
MID_DT=400 run sq354 16
grep -v "^\[cwr\]\|Warn\|warmup" $out
