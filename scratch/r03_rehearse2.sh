#!/bin/bash
# N = 2 rehearsal on one GPU: chained ranks with and without the overlapped (interior / cut lists) exchange, two runs each
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_rehearsal2.txt; : > $out
for rep in 1 2; do for no in 0 1; do
if [ $no = 1 ]; then export CWR_NO_OVERLAP=1; else unset CWR_NO_OVERLAP; fi
CWR_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so CWR_BENCH_DEVICE=0 timeout -k 10 500 python bench.py --gpus 2 --steps 12 --warmup 3 --windows 2 > /tmp/b.json 2> /tmp/b.err; rc=$?
python - $no $rc <<'PY' | tee -a $out
import json, sys
d = json.load(open('/tmp/b.json'))
it = d['solver']['iterations_per_step']
print(f"--gpus 2 CWR_NO_OVERLAP={sys.argv[1]} rc={sys.argv[2]}: sweeps {[i['sweeps'] for i in it]}, exchanges {[i['exchanges'] for i in it]}, overlapped {[i['overlapped'] for i in it]}, checks {[i['checks'] for i in it]}, {d['ms_per_step']} ms/step (shared GPU)")
PY
done; done
