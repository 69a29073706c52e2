#!/bin/bash
# `python bench.py --gpus N` (self-launched ranks) on ONE GPU through the stream-asynchronous RCCL stand-in: functional rehearsal of
# the N > 1 path (the ranks share the GPU: the rates mean nothing; sweeps, exchanges and checks per step do)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_rehearsal.txt; : > $out
for N in 2 4; do for nc in 0 1; do
if [ $nc = 1 ]; then export CWR_NO_CHAINS=1; else unset CWR_NO_CHAINS; fi
CWR_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so CWR_BENCH_DEVICE=0 timeout -k 10 500 python bench.py --gpus $N --steps 6 --warmup 2 --windows 2 > /tmp/b.json 2> /tmp/b.err; rc=$?
python - $N $nc $rc <<'PY' | tee -a $out
import json, sys
d = json.load(open('/tmp/b.json'))
it = d['solver']['iterations_per_step']
print(f"--gpus {sys.argv[1]} CWR_NO_CHAINS={sys.argv[2]} rc={sys.argv[3]}: n_gpus {d['n_gpus']}, {d['config']['partition']}, chained {d['solver']['chained_passes']}, "
      f"sweeps {[i['sweeps'] for i in it]}, exchanges {[i['exchanges'] for i in it]}, overlapped {[i['overlapped'] for i in it]}, checks {[i['checks'] for i in it]}, "
      f"{d['ms_per_step']} ms/step on the shared GPU")
PY
done; done
