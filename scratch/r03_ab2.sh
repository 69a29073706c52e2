#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03n_ab.txt; : > $out
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --windows 3 > /tmp/ab.json 2> /tmp/ab.err || { echo "$label FAILED" | tee -a $out; tail -3 /tmp/ab.err; return; }
  python - "$label" <<'PY' | tee -a $out
import json, sys
d = json.load(open('/tmp/ab.json')); r = d['roofline']
sw = [i['sweeps'] for i in d['solver']['iterations_per_step']]
print(f"{sys.argv[1]:34s} {d['value']:8.1f} Mcu/s {d['ms_per_step']:.3f} ms  pass {r['avg_launch_us']:.1f} us  frac {r['frac']:.3f}  sweeps {min(sw)}-{max(sw)}  read {(r['traffic_read'] or 0)/1e6:.1f} MB written {(r['traffic_written'] or 0)/1e6:.1f} MB ({(r['traffic_source'] or '')[:12]})")
PY
}
run "pingpong (round 2)"            CWR_NO_CHAINS=1
run "chains lanes, interleaved"     CWR_CHAIN_REUSE=0
run "chains lanes, column reuse"    CWR_CHAIN_REUSE=1
run "chains lanes, reuse, x3"       CWR_CHAIN_REUSE=1 CWR_LOCAL_REPS=3
