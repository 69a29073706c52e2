#!/bin/bash
# copy the outputs of scratch/r04_final.sh (gpurun_out/final4) into profiles/ as the set named $1 (default r04_s)
P=${1:-r04_s}; F=gpurun_out/final4
cp $F/bench_K16.json profiles/${P}_bench_line.json; cp $F/bench_K1.json profiles/${P}_bench_line_K1.json
cp $F/bench_K16_deterministic.json profiles/${P}_bench_line_deterministic.json
cp $F/bench_K16_under_rocprof.json profiles/${P}_bench_line_under_rocprof_K16.json; cp $F/bench_K1_under_rocprof.json profiles/${P}_bench_line_under_rocprof_K1.json
cat $F/config5.txt $F/ohio_like.txt > profiles/${P}_config5_and_small_meshes.txt
grep -v "\[warmup\]" $F/small_engines.txt > profiles/${P}_small_engines_r03_vs_r04.txt
cp $F/stiff.txt profiles/${P}_stiff.txt
cp $F/kernel_stats_bench_K16.csv profiles/${P}_kernel_stats_bench_K16.csv; cp $F/kernel_stats_bench_K1.csv profiles/${P}_kernel_stats_bench_K1.csv
cp $F/pmc_raw_summary.txt profiles/${P}_pmc_raw_summary.txt
python - $P <<'PY'
import json, sys, csv
P = sys.argv[1]
p = 'profiles/pmc_traffic.json'
d = json.load(open(p))
b = json.load(open(f'profiles/{P}_bench_line.json'))['roofline']; b1 = json.load(open(f'profiles/{P}_bench_line_K1.json'))['roofline']
if b.get('traffic_read') and b1.get('traffic_read'):
    d['bench_merged_1m'] = {'16': {'read': b['traffic_read'], 'written': b['traffic_written']}, '1': {'read': b1['traffic_read'], 'written': b1['traffic_written']}}
    json.dump(d, open(p, 'w'), indent=1)
for f in ['bench_line', 'bench_line_K1', 'bench_line_deterministic']:
    d = json.load(open(f'profiles/{P}_{f}.json')); r = d['roofline']
    print(f, d['value'], d['ms_per_step'], d['windows'], r['avg_launch_us'], r['frac'], r.get('traffic_read'), r.get('traffic_written'), r['achieved'], r.get('frac_kernel_bytes_read'), r.get('achieved_read_plus_write'), (d.get('cpu_baseline') or {}).get('value'), [i['sweeps'] for i in d['solver']['iterations_per_step']][-6:])
for K in (16, 1):
    rows = list(csv.DictReader(open(f'profiles/{P}_kernel_stats_bench_K{K}.csv')))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(K, [(r['Name'][5:28], r['Calls'], round(float(r['TotalDurationNs']) / tot * 100, 1), round(float(r['AverageNs']) / 1e3, 1)) for r in rows[:8]])
PY
