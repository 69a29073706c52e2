#!/bin/bash
# persistent grid of the tiled pass, same box: all resident slots (CWR_TCL_BALANCE=0) vs the fewest blocks with the same number
# of rounds (default), and blocks per CU capped
for r in 1 2; do
  for K in 1 16 8 4; do
    scratch/r02_quick.sh "K$K full grid            r$r" CWR_TCL_BALANCE=0 -- --constituents $K
    scratch/r02_quick.sh "K$K balanced grid        r$r" -- --constituents $K
    scratch/r02_quick.sh "K$K balanced, <=3 per CU r$r" CWR_TCL_BLOCKS_PER_CU=3 -- --constituents $K
    scratch/r02_quick.sh "K$K balanced, <=4 per CU r$r" CWR_TCL_BLOCKS_PER_CU=4 -- --constituents $K
  done
done
