#!/bin/bash
# full GPU suite + bench line (+ optional extra args for bench)
set -o pipefail
export TMPDIR=/tmp
tag=${1:-r03d}
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${tag}_tests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_bench.json'))
print(d['value'], d['ms_per_step'], d['windows']['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], [i['sweeps'] for i in d['solver']['iterations_per_step']])
PY
