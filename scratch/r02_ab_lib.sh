#!/bin/bash
# usage: scratch/r02_ab_lib.sh <libB.so> K...   -- the in-tree library (A) and libB (B) alternate on the same box, two rounds per K
B=$PWD/$1; shift
for K in "$@"; do
  for r in 1 2; do
    scratch/r02_quick.sh "K$K A (in-tree) r$r" -- --constituents $K --no-pmc
    scratch/r02_quick.sh "K$K B ($(basename $B)) r$r" CWR_TRANSPORT_LIB=$B -- --constituents $K --no-pmc
  done
done
