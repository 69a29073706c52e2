#!/bin/bash
# round 4: lane width (CWR_LANE_WIDTH_SCALE x 16 median centre distances) on the bench workload
out=gpurun_out/r04am_lane_width.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for sc in 1.0 0.94 0.97 1.03 1.06 1.0; do
    CWR_LANE_WIDTH_SCALE=$sc python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/r04am_tmp.json 2>gpurun_out/r04am_err.log || { tail -3 gpurun_out/r04am_err.log; exit 1; }
    python - $sc <<'PY' >> gpurun_out/r04am_lane_width.txt
import json, sys
d = json.loads(open('gpurun_out/r04am_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"scale {sys.argv[1]:5s}: {d['value']:8.1f} Mcell-upd/s {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
done
cat $out
