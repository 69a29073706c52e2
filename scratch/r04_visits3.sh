#!/bin/bash
# K = 16: passes per launch (default rule) vs one (CWR_VISITS=1) on other meshes / time steps / sizes
run() { local v=$1; shift
  CWR_VISITS=$v python bench.py --steps 12 --warmup 4 --windows 3 --no-cpu-baseline --no-pmc "$@" 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print('CWR_VISITS=%-2s %-34s %8.1f Mcell-upd/s %7.3f ms/step  windows %s  launches/step %5.1f  sweeps %s' % ('$v', '$*', d['value'], d['ms_per_step'], d['windows']['ms_per_step'], r['launches_timed']/d['steps'], [i['sweeps'] for i in d['solver']['iterations_per_step']][-3:]))"
}
for a in "--mesh quad" "--dt 20" "--dt 80" "--dt 160" "--nx 700 --ny 700" "--nx 1500 --ny 1000" "--nx 2000 --ny 2000 --steps 6 --warmup 3" "--diffusion 5.0" "--seed 9"; do for v in 1 8; do run $v $a; done; done
