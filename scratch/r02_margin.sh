#!/bin/bash
for K in 16 1; do
  CWR_TWO_CLOSING=1 python scratch/sweep_margin.py $K
  for m in 0 1 2; do CWR_SWEEP_MARGIN=$m python scratch/sweep_margin.py $K; done
done
