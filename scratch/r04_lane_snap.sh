#!/bin/bash
# round 4: lane boundaries snapped to the gaps between rows of cells (CWR_LANE_SNAP) x lane width (64 / CWR_LANE_LEN cells), K = 16
out=gpurun_out/r04bf_lane_snap.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for ll in 4 3.2 3 2.8 2.67 2.5; do
  for snap in 0 1; do
    CWR_LANE_SNAP=$snap CWR_LANE_LEN=$ll python bench.py --steps 20 --warmup 5 ${ARGS:-} --no-cpu-baseline --no-pmc > gpurun_out/r04bf_tmp.json 2>gpurun_out/r04bf_err.log || { tail -3 gpurun_out/r04bf_err.log; exit 1; }
    python - $snap $ll <<'PY' >> gpurun_out/r04bf_lane_snap.txt
import json, sys
d = json.loads(open('gpurun_out/r04bf_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"snap {sys.argv[1]} lane_len={sys.argv[2]:5s} (width {64/float(sys.argv[2]):5.2f}): {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
  done
done
cat $out
