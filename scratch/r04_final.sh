#!/bin/bash
# Final measurements of round 4 (one GPU box): bench lines, rocprofv3 kernel stats of the same command, PMC passes, config 5,
# Ohio-sized meshes, the stiff regime, engines below the chain threshold before / after.  Output: gpurun_out/final4 ->
# scratch/r04_install_final.sh copies the set into profiles/.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/final4; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_K16.json 2> $O/bench_K16.err; echo "bench K16 rc=$?"
python bench.py --steps 20 --warmup 5 --constituents 1 > $O/bench_K1.json 2> $O/bench_K1.err; echo "bench K1 rc=$?"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --deterministic > $O/bench_K16_deterministic.json 2>/dev/null; echo "bench deterministic rc=$?"
for K in 16 1; do
  rm -rf /tmp/prof_$K
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$K -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --constituents $K > $O/bench_K${K}_under_rocprof.json 2> /tmp/prof_$K.err
  cp $(find /tmp/prof_$K -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bench_K${K}.csv
done
for c in FETCH_SIZE "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  for K in 16 1; do
    d=/tmp/pmc_${K}_$(echo $c | cut -c1-5); rm -rf $d
    rocprofv3 --pmc $c --output-format csv -d $d -o pmc -- python3 scratch/pmc_target.py merged $K > /dev/null 2> $d.err
    echo "== K=$K counters: $c" >> $O/pmc_raw_summary.txt
    python scratch/pmc_summarize.py $d k_ >> $O/pmc_raw_summary.txt
  done
done
python scratch/config5.py > $O/config5.txt 2>&1
python tests/models/ohio_like.py > $O/ohio_like.txt 2>&1
: > $O/stiff.txt
python scratch/r03_stiff.py 16 6 40 pingpong 2 chains auto 2>&1 | grep -v Warn >> $O/stiff.txt
python scratch/r03_stiff.py 16 3 400 pingpong 2 chains auto 2>&1 | grep -v Warn >> $O/stiff.txt
python scratch/r03_stiff.py 16 3 1000 pingpong 2 chains auto 2>&1 | grep -v Warn >> $O/stiff.txt
python scratch/r03_stiff.py 16 2 3600 chains auto 2>&1 | grep -v Warn >> $O/stiff.txt
python scratch/r03_stiff.py 1 8 40 pingpong auto chains auto 2>&1 | grep -v Warn >> $O/stiff.txt
# engines below the chain threshold: round 3's choices (lanes, fixed applications) against round 4's (first combination = warm-up)
: > $O/small_engines.txt
C="warmup= r03=CWR_TILE_ORDER:lanes,CWR_NO_PP_REPS:1 r04="
for cs in "sq354 16" "sq245 16" "sq354 1" "band200x50 12" "band160x50 12" "band160x50 1"; do
  timeout -k 10 300 python scratch/r04_small.py $cs $C >> $O/small_engines.txt 2>&1
done
MID_DT=400 timeout -k 10 300 python scratch/r04_small.py sq354 16 $C >> $O/small_engines.txt 2>&1
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['windows']['ms_per_step'], r['avg_launch_us'], r['frac'], r.get('traffic_read'), r.get('traffic_written'), [i['sweeps'] for i in d['solver']['iterations_per_step']][-4:], (d.get('cpu_baseline') or {}).get('value'))
PY
done
tail -2 $O/config5.txt; tail -5 $O/ohio_like.txt; cat $O/stiff.txt; grep -v warmup $O/small_engines.txt
