#!/bin/bash
for K in 1 4 8 16; do
    scratch/r02_quick.sh "K$K static            " CWR_TCL_DYNAMIC=0 -- --constituents $K --no-pmc
    scratch/r02_quick.sh "K$K dynamic           " -- --constituents $K --no-pmc
    scratch/r02_quick.sh "K$K dynamic <=4 per CU" CWR_TCL_BLOCKS_PER_CU=4 -- --constituents $K --no-pmc
done
