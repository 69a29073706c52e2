#!/bin/bash
# usage: scratch/r02_prof.sh <tag> [bench args...]   -> gpurun_out/<tag>_*.{json,err,csv}
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
CWR_VERBOSE=1 python bench.py --no-pmc --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
grep '\[cwr\]' gpurun_out/${tag}_bench.err | sort | uniq -c > gpurun_out/${tag}_verbose.txt
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o run -- python3 bench.py --no-pmc --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/${tag}_bench_under_rocprof.json 2> /tmp/prof_$tag.err
find /tmp/prof_$tag -type f | head -20
f=$(find /tmp/prof_$tag -name '*kernel_stats*' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv && head -14 gpurun_out/${tag}_kernel_stats.csv | cut -c1-230
cat gpurun_out/${tag}_verbose.txt
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_bench.json'))
print('${tag}', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], [i['sweeps'] for i in d['solver']['iterations_per_step']])
PY
