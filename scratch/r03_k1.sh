#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03p_narrow_K.txt; : > $out
for K in 1 2 4; do
  timeout -k 10 300 python scratch/r03_stiff.py $K 8 40 pingpong auto chains auto chains 3 2>&1 | grep -v Warn | tee -a $out
done
timeout -k 10 300 python scratch/r03_stiff.py 16 8 40 chains auto 2>&1 | grep -v Warn | tee -a $out
