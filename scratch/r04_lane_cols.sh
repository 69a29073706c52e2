#!/bin/bash
out=gpurun_out/r04at_lane_cols.txt; : > $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for env in "CWR_LANE_COLUMNS=0" "CWR_LANE_COLUMNS=1" "CWR_LANE_COLUMNS=0.5" "CWR_LANE_COLUMNS=2" "CWR_LANE_COLUMNS=0" "CWR_LANE_COLUMNS=1"; do
    env $env python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/r04at_tmp.json 2>gpurun_out/r04at_err.log || { tail -3 gpurun_out/r04at_err.log; exit 1; }
    python - "$env" <<'PY' >> gpurun_out/r04at_lane_cols.txt
import json, sys
d = json.loads(open('gpurun_out/r04at_tmp.json').read().strip().splitlines()[-1])
it = d['solver']['iterations_per_step']; w = d['windows']['ms_per_step']
print(f"{sys.argv[1]:28s}: {d['value']:8.1f} Mcell-upd/s {d['ms_per_step']:7.3f} ms/step (windows {min(w):.3f}-{max(w):.3f})  pass {d['roofline']['avg_launch_us']:6.1f} us  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}")
PY
done
cat $out
