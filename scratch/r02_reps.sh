#!/bin/bash
# tile-local applications per pass (CWR_LOCAL_REPS) per K, same box
for K in ${KS:-1 2 4 8 16}; do
  scratch/r02_quick.sh "K$K default" -- --constituents $K --no-pmc
  for R in 2 3 4; do scratch/r02_quick.sh "K$K reps=$R" CWR_LOCAL_REPS=$R -- --constituents $K --no-pmc; done
done
