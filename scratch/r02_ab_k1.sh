#!/bin/bash
# K = 1 on the merged mesh, same box: default build (no work-item code) vs a -DCWR_WORK_ITEMS=1 build, split off / on
B=$PWD/scratch/libB_wi.so
for r in 1 2; do
  scratch/r02_quick.sh "K1 default build              r$r" -- --constituents 1
  scratch/r02_quick.sh "K1 work-item build, split off r$r" CWR_TRANSPORT_LIB=$B -- --constituents 1
  scratch/r02_quick.sh "K1 work-item build, split on  r$r" CWR_TRANSPORT_LIB=$B CWR_TCL_SPLIT=1 -- --constituents 1
done
