#!/bin/bash
# round 4: the 4 M-cell mesh (config 5's, without the reaction) at tile length 4 and 3, raw and smoothed lane boundaries
out=gpurun_out/r04bj_4m_len.txt; : > $out
for env in "CWR_LANE_LEN=4" "CWR_LANE_LEN=3" "CWR_LANE_LEN=4 CWR_LANE_SMOOTH=0" "CWR_LANE_LEN=3"; do
    env $env python bench.py --steps 10 --warmup 4 --nx 2052 --ny 2052 --seed 5 --no-cpu-baseline --no-pmc > gpurun_out/r04bj_tmp.json 2>gpurun_out/r04bj_err.log || { tail -3 gpurun_out/r04bj_err.log; exit 1; }
    python - "$env" <<'PY' >> gpurun_out/r04bj_4m_len.txt
import json, sys
d = json.loads(open('gpurun_out/r04bj_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"{sys.argv[1]:36s}: {d['config']['cells']} cells {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
done
cat $out
