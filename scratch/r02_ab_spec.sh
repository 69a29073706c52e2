#!/bin/bash
# preparing level t+1 beside the solve of level t (default) vs serial set-up (CWR_NO_SPEC_PREP=1), same box
for r in 1 2; do
  for K in 16 1 4; do
    scratch/r02_quick.sh "K$K serial set-up   r$r" CWR_NO_SPEC_PREP=1 -- --constituents $K --no-pmc
    scratch/r02_quick.sh "K$K prepared ahead  r$r" -- --constituents $K --no-pmc
  done
done
