#!/bin/bash
# Final measurements of round 6 (one GPU box).  Part A (default): bench lines (K = 16 with live PMC + CPU baseline, K = 1, deterministic,
# windowed flow field), rocprofv3 kernel stats of the same bench command, PMC passes.  Part B (`bash tools/r06_final.sh B`): config 5,
# Ohio-sized meshes, the stiff regime, engines below / at the chain threshold with round 6's knobs on and off.
# Output: gpurun_out/final6 -> copy the set into profiles/r06_final_* by hand.
set -o pipefail
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/final6; mkdir -p $O
part=${1:-A}
show() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d['windows']['ms_per_step'], r['avg_launch_us'], r['frac'], r.get('traffic_read'), r.get('traffic_written'), [i['sweeps'] for i in d['solver']['iterations_per_step']][-4:], (d.get('cpu_baseline') or {}).get('value'), d['config'].get('flow_field'))
    except Exception as ex:
        print(f, 'unreadable:', ex)
PY
}
if [ "$part" = A ]; then
  python bench.py --steps 20 --warmup 5 > $O/bench_K16.json 2> $O/bench_K16.err; echo "bench K16 rc=$?"
  python bench.py --steps 20 --warmup 5 --constituents 1 --no-cpu-baseline > $O/bench_K1.json 2> $O/bench_K1.err; echo "bench K1 rc=$?"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --deterministic > $O/bench_K16_deterministic.json 2>/dev/null; echo "bench deterministic rc=$?"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --flow-window 8 > $O/bench_K16_window8.json 2> $O/bench_K16_window8.err; echo "bench window rc=$?"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $O/bench_K16_again.json 2>/dev/null; echo "bench again rc=$?"
  CWR_EW_SPLIT=0 CWR_BOUND_WARM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $O/bench_K16_round5_rules.json 2>/dev/null; echo "bench round-5 rules rc=$?"
  for K in 16 1; do
    rm -rf /tmp/prof_$K
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$K -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --constituents $K > $O/bench_K${K}_under_rocprof.json 2> /tmp/prof_$K.err
    cp $(find /tmp/prof_$K -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bench_K${K}.csv
    python3 tools/trace_budget.py $(find /tmp/prof_$K -name '*kernel_trace.csv' | head -1) --steps 20 --label "bench.py K=$K (1 M cells), the 20 event-timed replay steps" > $O/step_budget_bench_K${K}.txt 2>&1
  done
  : > $O/pmc_raw_summary.txt
  for c in FETCH_SIZE "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    for K in 16 1; do
      d=/tmp/pmc_${K}_$(echo $c | cut -c1-5); rm -rf $d
      rocprofv3 --pmc $c --output-format csv -d $d -o pmc -- python3 tools/pmc_target.py merged $K > /dev/null 2> $d.err
      echo "== K=$K counters: $c" >> $O/pmc_raw_summary.txt
      python tools/pmc_summarize.py $d k_ >> $O/pmc_raw_summary.txt
    done
  done
  show $O/bench_*.json
else
  python tools/config5.py > $O/config5.txt 2>&1
  python tests/models/ohio_like.py > $O/ohio_like.txt 2>&1
  : > $O/stiff.txt
  python tools/r03_stiff.py 16 6 40 pingpong 2 chains auto 2>&1 | grep -v Warn >> $O/stiff.txt
  python tools/r03_stiff.py 16 3 400 pingpong 2 chains auto 2>&1 | grep -v Warn >> $O/stiff.txt
  python tools/r03_stiff.py 16 3 1000 pingpong 2 chains auto 2>&1 | grep -v Warn >> $O/stiff.txt
  python tools/r03_stiff.py 16 2 3600 chains auto 2>&1 | grep -v Warn >> $O/stiff.txt
  python tools/r03_stiff.py 1 8 40 pingpong auto chains auto 2>&1 | grep -v Warn >> $O/stiff.txt
  : > $O/small_engines.txt
  C="warmup= r05=CWR_EW_SPLIT:0,CWR_BOUND_WARM:0 r06="
  for cs in "sq354 16" "sq245 16" "sq354 1" "band200x50 12" "band160x50 12" "band160x50 1"; do
    timeout -k 10 300 python tools/r04_small.py $cs $C >> $O/small_engines.txt 2>&1
  done
  MID_DT=400 timeout -k 10 300 python tools/r04_small.py sq354 16 $C >> $O/small_engines.txt 2>&1
  tail -2 $O/config5.txt; tail -5 $O/ohio_like.txt; cat $O/stiff.txt; grep -v "warmup\|^\[cwr\]\|Warn" $O/small_engines.txt
fi
