#!/bin/bash
# round 4: tile length along the flow (CWR_LANE_LEN; lanes are 64 / it cells wide) in the stiff regime
out=gpurun_out/r04af_lane_len_stiff.txt
: > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for dt in 400 1000; do
  for ll in 4 8 16 2; do
    CWR_LANE_LEN=$ll python bench.py --steps 6 --warmup 3 --dt $dt --no-cpu-baseline --no-pmc > gpurun_out/r04af_tmp.json 2>gpurun_out/r04af_err.log || { tail -3 gpurun_out/r04af_err.log; exit 1; }
    python - $dt $ll <<'PY' >> gpurun_out/r04af_lane_len_stiff.txt
import json, sys
d = json.loads(open('gpurun_out/r04af_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']
print(f"dt={sys.argv[1]:>5s} lane_len={sys.argv[2]:>2s}: {d['ms_per_step']:7.3f} ms/step  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}  x{d['solver']['tile_local_applications']}")
PY
  done
done
cat $out
