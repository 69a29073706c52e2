#!/bin/bash
# usage: scratch/r02_quick.sh <label> [ENV=val ...] -- [bench args]   one bench line summary
label=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" == "--" ] && shift
env "${envs[@]}" python bench.py --steps 8 --warmup 3 --no-cpu-baseline "$@" > /tmp/q.json 2> /tmp/q.err || { tail -5 /tmp/q.err; exit 1; }
python - "$label" <<'PY'
import json,sys
d=json.load(open('/tmp/q.json'))
r=d['roofline']
print(f"{sys.argv[1]:40s} value {d['value']:8.1f}  ms/step {d['ms_per_step']:6.3f}  launch_us {r['avg_launch_us']:7.2f}  frac {r['frac']:.3f}  sweeps {[i['sweeps'] for i in d['solver']['iterations_per_step']][-3:]}")
PY
