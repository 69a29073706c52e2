#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03i_stiff.txt; : > $out
timeout -k 10 300 python scratch/r03_stiff.py 16 6 40 pingpong 2 chains 2 chains 3 2>&1 | grep -v Warn | tee -a $out
timeout -k 10 400 python scratch/r03_stiff.py 16 3 400 pingpong 2 pingpong 4 chains 2 chains 4 chains 6 chains 8 2>&1 | grep -v Warn | tee -a $out
timeout -k 10 500 python scratch/r03_stiff.py 16 3 1000 pingpong 2 pingpong 4 chains 4 chains 6 chains 8 chains 12 2>&1 | grep -v Warn | tee -a $out
