#!/bin/bash
# usage: scratch/r02_ab.sh <libB.so> [bench args]  -- alternate the in-tree library (A) and libB (B) on the same box, 3 rounds
B=$1; shift
for r in 1 2 3; do
  scratch/r02_quick.sh "A_round$r" -- "$@"
  scratch/r02_quick.sh "B_round$r" CWR_TRANSPORT_LIB=$PWD/$B -- "$@"
done
