#!/bin/bash
# one bench line per constituent count (merged 1 M-cell mesh, default settings) + the PMC detail of the chained pass at K = 16
export TMPDIR=/tmp
out=gpurun_out/r04p_per_K.txt; : > $out
for K in ${KS:-1 2 4 8 12 16 20 24 32}; do
  python bench.py --steps 10 --warmup 3 --windows 3 --no-cpu-baseline --no-pmc --constituents $K > /tmp/pk.json 2>/tmp/pk.err || { echo "K=$K FAILED" >> $out; tail -3 /tmp/pk.err >> $out; continue; }
  python - $K <<'PY' >> $out
import json, sys
d = json.load(open('/tmp/pk.json')); r = d['roofline']
it = d['solver']['iterations_per_step']
print(f"K={sys.argv[1]:>2s}: {d['value']:8.1f} Mcell-upd/s  {d['ms_per_step']:7.3f} ms/step  pass {r['avg_launch_us']:7.2f} us  frac {r['frac']:.3f}  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}  chained {d['solver']['chained_passes']}  x{d['solver']['tile_local_applications']}  numbering {d['config']['numbering']}")
PY
done
cat $out
bash scratch/r02_pmc_detail.sh 16 > /dev/null 2>&1; cp gpurun_out/pmcd/pmc_detail_K16.txt gpurun_out/r04p_pmc_detail_K16.txt 2>/dev/null; tail -30 gpurun_out/r04p_pmc_detail_K16.txt
