#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_behaviour.py -x -q -m gpu > gpurun_out/r03b_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03b_tests.log
timeout -k 10 500 python scratch/r03_chain.py bench 16 40 8 2 3 4 2>&1 | grep -v Warning | tee gpurun_out/r03b_chain_bench16.txt
timeout -k 10 300 python scratch/r03_chain.py bench 1 40 8 2 3 2>&1 | grep -v Warning | tee gpurun_out/r03b_chain_bench1.txt
timeout -k 10 500 python scratch/r03_chain.py bench 16 400 3 2 4 6 2>&1 | grep -v Warning | tee gpurun_out/r03b_chain_cfl25.txt
