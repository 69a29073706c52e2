#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/r03u_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r03u_tests.log
for N in 2; do
for nc in 0 1; do
if [ $nc = 1 ]; then export CWR_NO_CHAINS=1; else unset CWR_NO_CHAINS; fi
CWR_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so CWR_BENCH_DEVICE=0 timeout -k 10 500 python bench.py --gpus $N --steps 6 --warmup 2 --windows 2 > gpurun_out/r03u_bench_N${N}_$nc.json 2> gpurun_out/r03u_bench_N${N}_$nc.err; echo "N=$N no_chains=$nc rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/r03u_bench_N${N}_$nc.json'))
print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['partition'], [i['sweeps'] for i in d['solver']['iterations_per_step']])
PY
done; done
