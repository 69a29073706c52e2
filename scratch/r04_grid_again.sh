#!/bin/bash
# round 4, new numbering: persistent grid of the chained pass (blocks per CU -> list length) once more
out=gpurun_out/r04br_grid.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for env in "CWR_DUMMY=1" "CWR_TCL_BLOCKS_PER_CU=3" "CWR_TCL_BLOCKS_PER_CU=2" "CWR_TCL_GRID=896" "CWR_CHAIN_REFRESH=16" "CWR_CHAIN_REFRESH=1" "CWR_DUMMY=1"; do
    env $env python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/r04br_tmp.json 2>gpurun_out/r04br_err.log || { tail -3 gpurun_out/r04br_err.log; exit 1; }
    python - "$env" <<'PY' >> gpurun_out/r04br_grid.txt
import json, sys
d = json.loads(open('gpurun_out/r04br_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"{sys.argv[1].replace('CWR_DUMMY=1','default'):26s}: {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
done
cat $out
