#!/bin/bash
# internal face order (faces sorted by their smaller cell) at narrow K, same box
for K in 1 2 4; do
  for r in 1 2; do
    scratch/r02_quick.sh "K$K reference face order r$r" -- --constituents $K --no-pmc
    scratch/r02_quick.sh "K$K internal face order  r$r" CWR_FACE_ORDER_MIN_K=1 -- --constituents $K --no-pmc
  done
done
