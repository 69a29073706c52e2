#!/bin/bash
timeout -k 10 400 python -m pytest tests/test_gpu_multirank.py -k "eight" -x -q > gpurun_out/r04h_eight.log 2>&1; echo "eight rc=$?"; grep -E "mock_rccl|passed|failed|Error" gpurun_out/r04h_eight.log | cut -c1-400 | tail -12
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_multirank.py::test_eight_ranks_through_the_stand_in > gpurun_out/r04h_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r04h_tests.log
