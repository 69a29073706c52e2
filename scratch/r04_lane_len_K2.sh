#!/bin/bash
out=gpurun_out/r04be_lane_len_K2.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for K in 16 12 8 4 1; do
  for ll in 3 2.67 2.5 2.8 3; do
    CWR_LANE_LEN=$ll python bench.py --steps 20 --warmup 5 --constituents $K --no-cpu-baseline --no-pmc > gpurun_out/r04be_tmp.json 2>gpurun_out/r04be_err.log || { tail -3 gpurun_out/r04be_err.log; exit 1; }
    python - $K $ll <<'PY' >> gpurun_out/r04be_lane_len_K2.txt
import json, sys
d = json.loads(open('gpurun_out/r04be_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"K={sys.argv[1]:>2s} lane_len={sys.argv[2]:5s}: {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
  done
done
cat $out
