#!/bin/bash
# passes per launch of the chained in-place pass (CWR_VISITS): same-box A/B on the bench workload and neighbours
run() { # label, visits, bench args...
  local v=$1; shift
  CWR_VISITS=$v python bench.py --steps 20 --warmup 5 --windows 3 --no-cpu-baseline --no-pmc "$@" 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print('visits %2s %-28s %8.1f Mcell-upd/s %7.3f ms/step  windows %s  pass-kernel total %7.1f us/step  sweeps %s' % ('$v', '$*', d['value'], d['ms_per_step'], d['windows']['ms_per_step'], r['avg_launch_us']*r['launches_timed']/d['steps'], [i['sweeps'] for i in d['solver']['iterations_per_step']][-3:]))"
}
for v in 1 6 8 12 16 1 8; do run $v; done
for v in 1 8 4; do run $v --constituents 1; done
for v in 1 8 16; do run $v --dt 400 --steps 6 --warmup 2; done
for v in 1 8; do run $v --constituents 4; done
