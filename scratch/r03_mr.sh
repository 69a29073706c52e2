#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_chains.py -x -q -m gpu > gpurun_out/r03e_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r03e_tests.log
