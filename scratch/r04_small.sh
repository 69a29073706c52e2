#!/bin/bash
# same-box sweep over the knobs that already exist (round 4, task 1b)
set -o pipefail
out=gpurun_out/r04a_small.txt; : > $out
C="hilbert=CWR_TILE_ORDER:hilbert lanes=CWR_TILE_ORDER:lanes chain1=CWR_TILE_ORDER:lanes,CWR_CHAIN_MIN_TILES:1 chain1_b2=CWR_TILE_ORDER:lanes,CWR_CHAIN_MIN_TILES:1,CWR_TCL_BLOCKS_PER_CU:2 chain1_b1=CWR_TILE_ORDER:lanes,CWR_CHAIN_MIN_TILES:1,CWR_TCL_BLOCKS_PER_CU:1 chain1_r4=CWR_TILE_ORDER:lanes,CWR_CHAIN_MIN_TILES:1,CWR_LOCAL_REPS:4 chain1_b2_r4=CWR_TILE_ORDER:lanes,CWR_CHAIN_MIN_TILES:1,CWR_TCL_BLOCKS_PER_CU:2,CWR_LOCAL_REPS:4 pp_r4=CWR_TILE_ORDER:hilbert,CWR_LOCAL_REPS:4"
for cs in "sq354 16" "sq245 16" "sq354 1" "band200x50 12" "band160x50 12"; do
  timeout -k 10 240 python scratch/r04_small.py $cs $C >> $out 2>&1 || echo "FAILED $cs rc=$?" >> $out
done
tail -50 $out
