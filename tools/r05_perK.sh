#!/bin/bash
# VERDICT r04 task 5: one bench line per constituent count on the 1 M-cell bench mesh -- every K from 1 to 16 (and 20 .. 32 in
# steps), the padded counts also with CWR_K_PAD=0 (the caller's K as it is: round 4's behaviour).  Usage: bash tools/r05_perK.sh <tag>
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05i}
out=gpurun_out/${tag}_per_K.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
line() {  # K label env...
  local K=$1 label=$2; shift 2
  env "$@" python bench.py --steps 10 --warmup 3 --windows 3 --no-cpu-baseline --no-pmc --constituents $K > /tmp/pk.json 2>/tmp/pk.err || { echo "K=$K $label FAILED" >> $out; tail -3 /tmp/pk.err >> $out; return; }
  python - $K "$label" <<'PY' >> $out
import json, sys
d = json.load(open('/tmp/pk.json')); r = d['roofline']
it = d['solver']['iterations_per_step']
print(f"K={sys.argv[1]:>2s} {sys.argv[2]:9s}: {d['value']:8.1f} Mcell-upd/s  {d['ms_per_step']:7.3f} ms/step  pass {r['avg_launch_us']:7.2f} us  frac {r['frac']:.3f}  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}  chained {d['solver']['chained_passes']} det {d['solver']['deterministic_chained_passes']}  x{d['solver']['tile_local_applications']}  {r['kernel'][:24]}")
PY
}
for K in ${KS:-1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 18 20 22 24 28 32}; do
  line $K default CWR_DUMMY=1
  case $K in 3|5|7|9|10|11|13|14|15|18|22) line $K native CWR_K_PAD=0;; esac
done
cat $out
