#!/bin/bash
# blocks of the persistent tiled pass (CWR_TCL_GRID caps them), same box
for K in 1 16; do
  CWR_VERBOSE=1 CWR_TCL_BALANCE=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --constituents $K 2>&1 >/dev/null | grep "tiled J"
  for G in 512 768 896 960 984 1024 1152 1280; do
    scratch/r02_quick.sh "K$K grid<=$G" CWR_TCL_BALANCE=0 CWR_TCL_GRID=$G -- --constituents $K --no-pmc
  done
done
