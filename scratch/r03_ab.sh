#!/bin/bash
# A/B of chained passes and tile aspect on the bench line (same box), with live PMC traffic.
# (The 'hilbert aspect' lines need ordering.flow_aligned_order / CWR_TILE_ASPECT of commit 7b1c... -- the stretched Hilbert curve was
# measured (profiles/r03_c, section A) and removed when the lane-major order replaced it.)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03g_ab.txt; : > $out
run() { # label, env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --windows 3 > /tmp/ab.json 2> /tmp/ab.err || { echo "$label FAILED" | tee -a $out; tail -3 /tmp/ab.err; return; }
  python - "$label" <<'PY' | tee -a $out
import json, sys
d = json.load(open('/tmp/ab.json')); r = d['roofline']
sw = [i['sweeps'] for i in d['solver']['iterations_per_step']]
print(f"{sys.argv[1]:34s} {d['value']:8.1f} Mcu/s {d['ms_per_step']:.3f} ms  pass {r['avg_launch_us']:.1f} us  frac {r['frac']:.3f}  sweeps {min(sw)}-{max(sw)}  read {(r['traffic_read'] or 0)/1e6:.1f} MB written {(r['traffic_written'] or 0)/1e6:.1f} MB ({(r['traffic_source'] or '')[:12]})")
PY
}
run "pingpong iso (round 2)"  CWR_NO_CHAINS=1
run "chains hilbert iso" CWR_TILE_ORDER=hilbert CWR_TILE_ASPECT=1
run "chains hilbert aspect 2" CWR_TILE_ORDER=hilbert CWR_TILE_ASPECT=2
run "chains hilbert aspect 2 reps 3" CWR_TILE_ORDER=hilbert CWR_TILE_ASPECT=2 CWR_LOCAL_REPS=3
run "chains hilbert aspect 1.5" CWR_TILE_ORDER=hilbert CWR_TILE_ASPECT=1.5
run "chains lanes 4x16"       CWR_TILE_ORDER=lanes
run "chains lanes reps 3"     CWR_TILE_ORDER=lanes CWR_LOCAL_REPS=3
