#!/bin/bash
# round 4: `python bench.py --gpus N` (self-launched ranks) on ONE GPU through the stand-in, N = 4 and 6 (the box allows six GPU processes):
# 167 k cells per rank at N = 6 -- 2.5 tiles per block, chained since the threshold fell to 1.75.  The rates mean nothing (shared GPU).
set -o pipefail
export TMPDIR=/tmp
(cd tests/mock_rccl && make -s 2>/dev/null || true)
out=gpurun_out/r04bb_rehearsal6.txt; : > $out
for N in 4 6; do for mode in default min3; do
if [ $mode = min3 ]; then export CWR_CHAIN_MIN_TILES=3 CWR_TILE_ORDER=hilbert; else unset CWR_CHAIN_MIN_TILES CWR_TILE_ORDER; fi
CWR_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so CWR_BENCH_DEVICE=0 timeout -k 10 600 python bench.py --gpus $N --steps 6 --warmup 2 --windows 2 > /tmp/b.json 2> /tmp/b.err; rc=$?
[ $rc = 0 ] || tail -5 /tmp/b.err
python - $N $mode $rc <<'PY' | tee -a $out
import json, sys
d = json.load(open('/tmp/b.json'))
it = d['solver']['iterations_per_step']
print(f"--gpus {sys.argv[1]} {sys.argv[2]} rc={sys.argv[3]}: n_gpus {d['n_gpus']}, {d['config']['numbering']}, {d['config']['partition']}, chained {d['solver']['chained_passes']}, "
      f"sweeps {[i['sweeps'] for i in it]}, exchanges {[i['exchanges'] for i in it]}, overlapped {[i['overlapped'] for i in it]}, checks {[i['checks'] for i in it]}, "
      f"{d['ms_per_step']} ms/step on the shared GPU")
PY
done; done
