#!/bin/bash
# one bench line per constituent count with the final code (1 M-cell bench mesh, default settings)
export TMPDIR=/tmp
out=gpurun_out/r04bq_per_K_final.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for K in ${KS:-1 2 3 4 5 6 8 10 12 16 20 24 28 32}; do
  python bench.py --steps 10 --warmup 3 --windows 3 --no-cpu-baseline --no-pmc --constituents $K > /tmp/pk.json 2>/tmp/pk.err || { echo "K=$K FAILED" >> $out; tail -3 /tmp/pk.err >> $out; continue; }
  python - $K <<'PY' >> $out
import json, sys
d = json.load(open('/tmp/pk.json')); r = d['roofline']
it = d['solver']['iterations_per_step']
print(f"K={sys.argv[1]:>2s}: {d['value']:8.1f} Mcell-upd/s  {d['ms_per_step']:7.3f} ms/step  pass {r['avg_launch_us']:7.2f} us  frac {r['frac']:.3f}  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}  chained {d['solver']['chained_passes']} det {d['solver']['deterministic_chained_passes']}  x{d['solver']['tile_local_applications']}  {r['kernel'][:24]}")
PY
done
cat $out
