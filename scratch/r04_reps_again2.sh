#!/bin/bash
out=gpurun_out/r04bl_reps2.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
run() { # args reps...
  a="$1"; shift
  for r in "$@"; do
    CWR_LOCAL_REPS=$r python bench.py $a --no-cpu-baseline --no-pmc > gpurun_out/r04bl_tmp.json 2>gpurun_out/r04bl_err.log || { tail -3 gpurun_out/r04bl_err.log; exit 1; }
    python - "$r" "$a" <<'PY' >> gpurun_out/r04bl_reps2.txt
import json, sys
d = json.loads(open('gpurun_out/r04bl_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"reps {sys.argv[1]} {sys.argv[2]:34s}: {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
  done
}
run "--dt 200 --steps 8 --warmup 3" 2 3 4
run "--dt 1000 --steps 4 --warmup 2" 3 4 5 6
run "--dt 3600 --steps 3 --warmup 2" 4 6 8
cat $out
