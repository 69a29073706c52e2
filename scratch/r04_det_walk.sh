#!/bin/bash
# round 4: deterministic steps walk the chain lists (ping-pong vectors, predecessor rows from LDS): A/B against tile order
out=gpurun_out/r04ad_det_walk.txt
: > $out
line() { python - "$1" "$2" <<'PY' >> gpurun_out/r04ad_det_walk.txt
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
sw = d['config'].get('iterations_last_step') or d['config'].get('iters')
print(f"{sys.argv[1]:44s} {d['value']:8.1f} Mcell-upd/s  {d['ms_per_step']:7.3f} ms/step  pass {d['roofline']['launch_us']:6.1f} us" if 'launch_us' in d['roofline'] else f"{sys.argv[1]:44s} {d['value']:8.1f} Mcell-upd/s  {d['ms_per_step']:7.3f} ms/step  achieved {d['roofline']['achieved']:7.1f} GB/s", ' sweeps', d.get('sweeps_per_step', d['config'].get('sweeps')))
PY
}
python bench.py --steps 20 --warmup 5 --deterministic --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for args in "" "--constituents 1" "--constituents 4" "--dt 400 --steps 6 --warmup 3" "--dt 1000 --steps 4 --warmup 2"; do
  for w in 1 0; do
    CWR_DET_WALK=$w python bench.py --steps 20 --warmup 5 $args --deterministic --no-cpu-baseline --no-pmc > gpurun_out/r04ad_tmp.json 2>gpurun_out/r04ad_err.log || { tail -3 gpurun_out/r04ad_err.log; exit 1; }
    line "det walk=$w $args" gpurun_out/r04ad_tmp.json
  done
  python bench.py --steps 20 --warmup 5 $args --no-cpu-baseline --no-pmc > gpurun_out/r04ad_tmp.json 2>gpurun_out/r04ad_err.log || exit 1
  line "in place (default) $args" gpurun_out/r04ad_tmp.json
done
cat $out
