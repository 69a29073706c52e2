#!/bin/bash
# round 4: the bench workload with straight lanes (default there) against lanes from channel_coordinates
out=gpurun_out/r04al_bench_lane_kind.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for kind in straight channel straight channel; do
  for args in "" "--constituents 1" "--dt 400 --steps 6 --warmup 3"; do
    CWR_LANE_KIND=$kind python bench.py --steps 20 --warmup 5 $args --no-cpu-baseline --no-pmc > gpurun_out/r04al_tmp.json 2>gpurun_out/r04al_err.log || { tail -3 gpurun_out/r04al_err.log; exit 1; }
    python - $kind "$args" <<'PY' >> gpurun_out/r04al_bench_lane_kind.txt
import json, sys
d = json.loads(open('gpurun_out/r04al_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"{sys.argv[1]:9s} {sys.argv[2]:34s}: {d['value']:8.1f} Mcell-upd/s {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
  done
done
cat $out
