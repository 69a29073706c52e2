#!/bin/bash
# round 4: tile-local applications per visit once more, with the new numbering (3-cell tiles, smoothed lanes)
out=gpurun_out/r04bk_reps.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for args in "" "--dt 100" "--dt 400 --steps 6 --warmup 3"; do
  for env in "CWR_VERBOSE=0" "CWR_LOCAL_REPS=2" "CWR_LOCAL_REPS=3" "CWR_LOCAL_REPS=4" "CWR_LOCAL_REPS=6"; do
    env ${env/CWR_VERBOSE=0/CWR_DUMMY=0} python bench.py --steps 20 --warmup 5 $args --no-cpu-baseline --no-pmc > gpurun_out/r04bk_tmp.json 2>gpurun_out/r04bk_err.log || { tail -3 gpurun_out/r04bk_err.log; exit 1; }
    python - "$env" "$args" <<'PY' >> gpurun_out/r04bk_reps.txt
import json, sys
d = json.loads(open('gpurun_out/r04bk_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"{sys.argv[1].replace('CWR_VERBOSE=0','auto'):18s} {sys.argv[2]:30s}: x{d['solver']['tile_local_applications']} {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
  done
done
cat $out
