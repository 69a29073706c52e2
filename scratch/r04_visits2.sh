#!/bin/bash
# passes per launch chosen from the list length (default) against one pass per launch (CWR_VISITS=1), alternating, per K
run() { local v=$1; shift
  CWR_VISITS=$v python bench.py --steps 20 --warmup 5 --windows 3 --no-cpu-baseline --no-pmc "$@" 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print('CWR_VISITS=%-2s %-22s %8.1f Mcell-upd/s %7.3f ms/step  windows %s  pass-kernel total %7.1f us/step in %5.1f launches  sweeps %s' % ('$v', '$*', d['value'], d['ms_per_step'], d['windows']['ms_per_step'], r['avg_launch_us']*r['launches_timed']/d['steps'], r['launches_timed']/d['steps'], [i['sweeps'] for i in d['solver']['iterations_per_step']][-3:]))"
}
for K in 16 12 8 20 32 4 1; do for v in 1 8 1 8; do run $v --constituents $K; done; done
