#!/bin/bash
for K in 1 2 4; do for r in 1 2; do
  scratch/r02_quick.sh "K$K default r$r" -- --constituents $K --no-pmc
  scratch/r02_quick.sh "K$K nt stream r$r" CWR_NT_STREAM=1 -- --constituents $K --no-pmc
done; done
