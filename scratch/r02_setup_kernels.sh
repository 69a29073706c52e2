#!/bin/bash
# per-kernel averages of the per-level set-up kernels (rocprofv3 --kernel-trace --stats over a short bench run)
export TMPDIR=/tmp
for K in 16 1; do
  rm -rf /tmp/su_$K
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/su_$K -o run -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-pmc --constituents $K > /tmp/su_$K.json 2>/dev/null
  python - $(find /tmp/su_$K -name '*kernel_stats.csv' | head -1) $K /tmp/su_$K.json <<'PY'
import csv,sys,json
d=json.load(open(sys.argv[3]))
print(f"K={sys.argv[2]}: {d['ms_per_step']} ms/step (under the profiler)")
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_prep_step','k_sq_numeric','k_rhs','k_mass_flux','k_sq_tiled','k_apply',"k_reduce")):
        print(f"   {r['Name'][:44]:44s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
