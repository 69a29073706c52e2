#!/bin/bash
# round 4: lane width in cells (16 x CWR_LANE_WIDTH_SCALE) with the smoothed boundaries, K = 16, CFL 2.5 and CFL 25
out=gpurun_out/r04bd_lane_width2.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for args in "" "--dt 400 --steps 6 --warmup 3"; do
  for sc in 1.0 1.125 1.25 1.3125 1.4 1.5 1.75 1.0; do
    CWR_LANE_WIDTH_SCALE=$sc python bench.py --steps 20 --warmup 5 $args --no-cpu-baseline --no-pmc > gpurun_out/r04bd_tmp.json 2>gpurun_out/r04bd_err.log || { tail -3 gpurun_out/r04bd_err.log; exit 1; }
    python - $sc "$args" <<'PY' >> gpurun_out/r04bd_lane_width2.txt
import json, sys
d = json.loads(open('gpurun_out/r04bd_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"width {16*float(sys.argv[1]):5.1f} cells {sys.argv[2]:30s}: {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
  done
done
cat $out
