#!/bin/bash
timeout -k 10 400 python -m pytest tests/test_gpu_multirank.py -k "eight" -x -q > gpurun_out/r04f_eight.log 2>&1; echo "eight rc=$?"; grep -E "mock_rccl|passed|failed|Error" gpurun_out/r04f_eight.log | cut -c1-400 | tail -12
timeout -k 10 600 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_parity.py tests/test_gpu_behaviour.py tests/test_gpu_chains.py -x -q > gpurun_out/r04f_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r04f_tests.log
bash scratch/r04_small2.sh gpurun_out/r04f_small.txt
