#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03o_stiff_reuse.txt; : > $out
timeout -k 10 300 python scratch/r03_stiff.py 16 6 40 pingpong 2 chains auto chains 3 2>&1 | grep -v Warn | tee -a $out
timeout -k 10 400 python scratch/r03_stiff.py 16 3 400 pingpong 2 chains 2 chains auto chains 6 2>&1 | grep -v Warn | tee -a $out
timeout -k 10 500 python scratch/r03_stiff.py 16 3 1000 pingpong 2 chains auto chains 6 2>&1 | grep -v Warn | tee -a $out
timeout -k 10 500 python scratch/r03_stiff.py 16 2 3600 pingpong 2 chains auto chains 8 2>&1 | grep -v Warn | tee -a $out
