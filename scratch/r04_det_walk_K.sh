#!/bin/bash
# round 4: deterministic chained passes against the in-place ones at other constituent counts
out=gpurun_out/r04ai_det_walk_K.txt
: > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for K in 1 2 4 8 12; do
  for mode in "--deterministic" ""; do
    python bench.py --steps 20 --warmup 5 --constituents $K $mode --no-cpu-baseline --no-pmc > gpurun_out/r04ai_tmp.json 2>gpurun_out/r04ai_err.log || { tail -3 gpurun_out/r04ai_err.log; exit 1; }
    python - $K "${mode:-in-place}" <<'PY' >> gpurun_out/r04ai_det_walk_K.txt
import json, sys
d = json.loads(open('gpurun_out/r04ai_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']
w = d['windows']['ms_per_step']
print(f"K={sys.argv[1]:>2s} {sys.argv[2]:16s}: {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
  done
done
cat $out
