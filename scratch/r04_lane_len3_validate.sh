#!/bin/bash
# round 4: tile length 3 (the new default) against 4 across regimes, same box
out=gpurun_out/r04bg_lane_len3.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for args in "" "--deterministic" "--dt 400 --steps 6 --warmup 3" "--dt 1000 --steps 4 --warmup 2" "--constituents 20" "--constituents 32" "--mesh quad"; do
  for ll in 4 3; do
    CWR_LANE_LEN=$ll python bench.py --steps 20 --warmup 5 $args --no-cpu-baseline --no-pmc > gpurun_out/r04bg_tmp.json 2>gpurun_out/r04bg_err.log || { tail -3 gpurun_out/r04bg_err.log; exit 1; }
    python - $ll "$args" <<'PY' >> gpurun_out/r04bg_lane_len3.txt
import json, sys
d = json.loads(open('gpurun_out/r04bg_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"lane_len={sys.argv[1]} {sys.argv[2]:32s}: {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
  done
done
C="warmup= len4=CWR_LANE_LEN:4 len3=CWR_LANE_LEN:3 len4b=CWR_LANE_LEN:4 len3b=CWR_LANE_LEN:3"
for cs in "sq354 16" "sq400 16" "bend1026x256@1.0 16" "sq700 1"; do timeout -k 10 400 python scratch/r04_small.py $cs $C 2>&1 | grep -v "^\[cwr\]\|Warn\|warmup" | cut -c1-180 >> $out; done
cat $out
