#!/bin/bash
# multi-visit launches with bounded-drift pacing: CWR_VISITS x CWR_PACE_LAG, per workload
run() { local v=$1 l=$2; shift 2
  CWR_VISITS=$v CWR_PACE_LAG=$l python bench.py --steps 12 --warmup 4 --windows 3 --no-cpu-baseline --no-pmc "$@" 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print('visits %-2s lag %s %-26s %8.1f Mcell-upd/s %7.3f ms/step  windows %s  launches/step %5.1f  sweeps %s' % ('$v', '$l', '$*', d['value'], d['ms_per_step'], d['windows']['ms_per_step'], r['launches_timed']/d['steps'], [i['sweeps'] for i in d['solver']['iterations_per_step']][-3:]))"
}
for a in "" "--constituents 12" "--constituents 4" "--mesh quad" "--dt 160" "--constituents 20"; do
  run 1 0 $a; run 8 0 $a; run 8 1 $a; run 8 2 $a
done
