#!/bin/bash
# one bench line per constituent count (merged 1 M-cell mesh, 8-step windows): ms per step, dominant kernel, frac, sweeps
export TMPDIR=/tmp
out=gpurun_out/r03_perK.txt; : > $out
for K in 1 2 3 4 6 8 12 16 20 24 32; do
  timeout -k 10 300 python bench.py --constituents $K --steps 8 --warmup 3 --windows 2 --no-cpu-baseline --no-pmc > /tmp/k.json 2>/dev/null || { echo "K=$K FAILED" | tee -a $out; continue; }
  python - $K <<'PY' | tee -a $out
import json, sys
d = json.load(open('/tmp/k.json')); r = d['roofline']; sw = [i['sweeps'] for i in d['solver']['iterations_per_step']]
print(f"K={int(sys.argv[1]):2d}: {d['ms_per_step']:7.3f} ms/step  {d['value']:8.1f} Mcell-upd/s  pass {r['avg_launch_us']:6.1f} us  frac {r['frac']:.3f}  sweeps {min(sw)}-{max(sw)}  x{d['solver']['tile_local_applications']}  chained {d['solver']['chained_passes']}")
PY
done
