#!/bin/bash
# 8-rank diagnostics, then the chain / parity tests with multi-visit launches, then the small-mesh sweep
timeout -k 10 300 python -m pytest tests/test_gpu_multirank.py -k "eight" -x -q > gpurun_out/r04e_eight.log 2>&1; echo "eight rc=$?"; grep -E "mock_rccl|passed|failed" gpurun_out/r04e_eight.log | cut -c1-600 | tail -20
timeout -k 10 500 python -m pytest tests/test_gpu_chains.py tests/test_gpu_parity.py tests/test_gpu_behaviour.py tests/test_gpu_robustness.py -x -q > gpurun_out/r04e_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r04e_tests.log
bash scratch/r04_small.sh gpurun_out/r04e_small.txt
