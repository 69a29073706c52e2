#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_outputs.py tests/test_gpu_robustness.py tests/test_gpu_behaviour.py -x -q -m gpu > gpurun_out/r03j_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03j_tests.log
timeout -k 10 300 python tests/models/ohio_like.py 2>&1 | grep -v Warn | tee gpurun_out/r03j_small_meshes.txt
