#!/bin/bash
# round 4: A/B of the inner loop of the tiled pass (scratch/ab/libcwr_loop2.so = -DCWR_LOOP2=1) against the default library, alternating
out=gpurun_out/r04bp_loop2.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for args in "" "--constituents 12" "--dt 400 --steps 6 --warmup 3"; do
  for lib in default loop2 default loop2; do
    if [ $lib = default ]; then unset CWR_TRANSPORT_LIB; else export CWR_TRANSPORT_LIB=$PWD/scratch/ab/libcwr_loop2.so; fi
    python bench.py --steps 20 --warmup 5 $args --no-cpu-baseline --no-pmc > gpurun_out/r04bp_tmp.json 2>gpurun_out/r04bp_err.log || { tail -3 gpurun_out/r04bp_err.log; exit 1; }
    python - $lib "$args" <<'PY' >> gpurun_out/r04bp_loop2.txt
import json, sys
d = json.loads(open('gpurun_out/r04bp_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"{sys.argv[1]:8s} {sys.argv[2]:30s}: {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
  done
done
cat $out
