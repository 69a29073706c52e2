#!/bin/bash
# round 4, after the smoothed lane boundaries: deterministic chained passes against the in-place ones
out=gpurun_out/r04au_det_gap.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for args in "--constituents 8" "--constituents 12" "--constituents 16" "--constituents 20" "--constituents 32" "--dt 400 --steps 6 --warmup 3" "--dt 1000 --steps 4 --warmup 2" "--dt 400 --steps 6 --warmup 3 --constituents 4"; do
  for mode in "--deterministic" ""; do
    CWR_DET_DEFAULT_K=0 python bench.py --steps 20 --warmup 5 $args $mode --no-cpu-baseline --no-pmc > gpurun_out/r04au_tmp.json 2>gpurun_out/r04au_err.log || { tail -3 gpurun_out/r04au_err.log; exit 1; }
    python - "$args" "${mode:-in place}" <<'PY' >> gpurun_out/r04au_det_gap.txt
import json, sys
d = json.loads(open('gpurun_out/r04au_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"{sys.argv[1]:48s} {sys.argv[2]:16s}: {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
  done
done
cat $out
