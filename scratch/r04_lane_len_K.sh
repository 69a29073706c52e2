#!/bin/bash
# round 4: tile length along the flow (CWR_LANE_LEN) at other constituent counts, with the smoothed lane boundaries
out=gpurun_out/r04bc_lane_len_K.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for K in ${KS:-1 4 12}; do
  for ll in ${LLS:-4 2 3 5 8 4}; do
    CWR_LANE_LEN=$ll python bench.py --steps 20 --warmup 5 --constituents $K --no-cpu-baseline --no-pmc > gpurun_out/r04bc_tmp.json 2>gpurun_out/r04bc_err.log || { tail -3 gpurun_out/r04bc_err.log; exit 1; }
    python - $K $ll <<'PY' >> gpurun_out/r04bc_lane_len_K.txt
import json, sys
d = json.loads(open('gpurun_out/r04bc_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"K={sys.argv[1]:>2s} lane_len={sys.argv[2]}: {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
  done
done
cat $out
