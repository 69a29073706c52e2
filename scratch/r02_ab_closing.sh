#!/bin/bash
# batch shape on one GPU, same box: even passes + two closing sweeps (CWR_TWO_CLOSING=1, round 1) vs any passes + one closing sweep
for r in 1 2; do
  for K in 16 1 8; do
    scratch/r02_quick.sh "K$K two closing sweeps r$r" CWR_TWO_CLOSING=1 -- --constituents $K --no-pmc
    scratch/r02_quick.sh "K$K one closing sweep  r$r" -- --constituents $K --no-pmc
  done
done
