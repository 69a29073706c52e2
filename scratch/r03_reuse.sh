#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_chains.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r03l_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03l_tests.log
out=gpurun_out/r03l_reuse.txt; : > $out
for K in 16 12 8; do
for reuse in 1 0; do
  echo "== K=$K CWR_CHAIN_REUSE=$reuse" | tee -a $out
  CWR_CHAIN_REUSE=$reuse timeout -k 10 300 python scratch/r03_stiff.py $K 8 40 chains auto 2>&1 | grep -v Warn | tee -a $out
done; done
CWR_CHAIN_REUSE=1 timeout -k 10 300 python scratch/r03_stiff.py 16 3 400 chains auto 2>&1 | grep -v Warn | tee -a $out
CWR_CHAIN_REUSE=0 timeout -k 10 300 python scratch/r03_stiff.py 16 3 400 chains auto 2>&1 | grep -v Warn | tee -a $out
