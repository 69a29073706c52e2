#!/bin/bash
# round 4: sweeps of margin on the first batch of a step (CWR_SWEEP_MARGIN), bench workload and a stiff step
out=gpurun_out/r04bo_margin.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for args in "" "--constituents 1" "--dt 400 --steps 6 --warmup 3"; do
  for m in 1 0 2 1 0; do
    CWR_SWEEP_MARGIN=$m python bench.py --steps 20 --warmup 5 $args --no-cpu-baseline --no-pmc > gpurun_out/r04bo_tmp.json 2>gpurun_out/r04bo_err.log || { tail -3 gpurun_out/r04bo_err.log; exit 1; }
    python - $m "$args" <<'PY' >> gpurun_out/r04bo_margin.txt
import json, sys
d = json.loads(open('gpurun_out/r04bo_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"margin {sys.argv[1]} {sys.argv[2]:30s}: {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  sweeps {[i['sweeps'] for i in it]}")
PY
  done
done
cat $out
