#!/bin/bash
# one bench line per constituent count, default settings (merged 1 M-cell mesh)
for K in ${KS:-1 2 3 4 6 8 12 16 32}; do
  scratch/r02_quick.sh "K$K" -- --constituents $K --no-pmc
done
