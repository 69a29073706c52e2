#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest "tests/test_gpu_multirank.py::test_partitioned_block_asynchronous_passes_keep_the_single_rank_sweep_count" -x -q -m gpu 2>&1 | tail -3
for N in 2 4; do
CWR_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so CWR_BENCH_DEVICE=0 timeout -k 10 500 python bench.py --gpus $N --steps 6 --warmup 2 --windows 2 > gpurun_out/r03t_bench_N$N.json 2> gpurun_out/r03t_bench_N$N.err; echo "N=$N rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/r03t_bench_N$N.json'))
print(d['n_gpus'], d['value'], d['ms_per_step'], d['windows']['ms_per_step'], d['config']['partition'], [i['sweeps'] for i in d['solver']['iterations_per_step']])
PY
done
