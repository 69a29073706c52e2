#!/bin/bash
# tile schedule of the tiled pass, same box: static strided (CWR_TCL_DYNAMIC=0) vs per-XCD counters (default)
for r in 1 2; do
  for K in 1 4 8 16; do
    scratch/r02_quick.sh "K$K static            r$r" CWR_TCL_DYNAMIC=0 -- --constituents $K --no-pmc
    scratch/r02_quick.sh "K$K dynamic           r$r" -- --constituents $K --no-pmc
    scratch/r02_quick.sh "K$K dynamic <=4 per CU r$r" CWR_TCL_BLOCKS_PER_CU=4 -- --constituents $K --no-pmc
  done
done
