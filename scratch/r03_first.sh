#!/bin/bash
# round 3, first GPU call: the new tests, the bench line, and the self-launched 2-rank rehearsal on one GPU (mock RCCL)
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/r03a_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r03a_tests.log
tail -5 gpurun_out/r03a_tests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; echo "bench rc=$?"
cut -c1-600 gpurun_out/r03a_bench.json
CWR_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so CWR_BENCH_DEVICE=0 timeout -k 10 400 python bench.py --gpus 2 --steps 4 --warmup 2 --nx 600 --ny 600 > gpurun_out/r03a_bench2.json 2> gpurun_out/r03a_bench2.err; echo "bench2 rc=$?"
cut -c1-400 gpurun_out/r03a_bench2.json; tail -3 gpurun_out/r03a_bench2.err
