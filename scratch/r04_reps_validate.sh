#!/bin/bash
# round 4: the new rule for the tile-local applications (2 / 3 / 4 by ||J||inf) against round 3's (2 / 4 / 6 / 8), other K
out=gpurun_out/r04bm_reps_validate.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
run() { a="$1"; shift
  for r in "$@"; do
    if [ $r = auto ]; then unset CWR_LOCAL_REPS; else export CWR_LOCAL_REPS=$r; fi
    python bench.py $a --no-cpu-baseline --no-pmc > gpurun_out/r04bm_tmp.json 2>gpurun_out/r04bm_err.log || { tail -3 gpurun_out/r04bm_err.log; exit 1; }
    python - "$r" "$a" <<'PY' >> gpurun_out/r04bm_reps_validate.txt
import json, sys
d = json.loads(open('gpurun_out/r04bm_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"reps {sys.argv[1]:4s} {sys.argv[2]:52s}: x{d['solver']['tile_local_applications']} {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
  done; unset CWR_LOCAL_REPS; }
run "--dt 400 --steps 6 --warmup 3 --constituents 1" auto 4 2
run "--dt 400 --steps 6 --warmup 3 --constituents 4" auto 4
run "--dt 1000 --steps 4 --warmup 2 --constituents 1" auto 6 3
run "--dt 1000 --steps 4 --warmup 2 --constituents 4" auto 6
run "--dt 400 --steps 6 --warmup 3 --deterministic" auto 4
run "--dt 1000 --steps 4 --warmup 2 --deterministic" auto 6
run "--dt 400 --steps 6 --warmup 3 --constituents 12" auto 4
cat $out
