#!/bin/bash
# round 4: lane boundaries cut in a smoothed across-coordinate (CWR_LANE_SMOOTH sweeps of neighbour averaging; 0 = the raw coordinate)
out=gpurun_out/r04aq_lane_smooth.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for args in "" "--constituents 1" "--constituents 4" "--constituents 12" "--dt 400 --steps 6 --warmup 3" "--deterministic"; do
  for sm in 0 32 0 32; do
    CWR_LANE_SMOOTH=$sm python bench.py --steps 20 --warmup 5 $args --no-cpu-baseline --no-pmc > gpurun_out/r04aq_tmp.json 2>gpurun_out/r04aq_err.log || { tail -3 gpurun_out/r04aq_err.log; exit 1; }
    python - $sm "$args" <<'PY' >> gpurun_out/r04aq_lane_smooth.txt
import json, sys
d = json.loads(open('gpurun_out/r04aq_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"smooth {sys.argv[1]:3s} {sys.argv[2]:32s}: {d['value']:8.1f} Mcell-upd/s {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
  done
done
cat $out
