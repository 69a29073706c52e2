#!/bin/bash
# VERDICT r04 task 4: the per-step upload of a windowed flow field, hidden behind the steps -- bench.py with all levels resident against
# bench.py --flow-window 8 (a ring of 8 levels, one level uploaded per step on the engine's flow stream from page-locked host arrays), same
# box, alternating.  Usage (GPU box): bash tools/r05_window.sh <tag>
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05w}
O=gpurun_out/${tag}_window; mkdir -p $O
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # warm the box
for rep in 1 2; do
  for K in 16 1; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --constituents $K > $O/resident_K${K}_$rep.json 2>/dev/null
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --constituents $K --flow-window 8 > $O/window8_K${K}_$rep.json 2> $O/window8_K${K}_$rep.err
  done
done
python - $O <<'PY' | tee gpurun_out/${tag}_window.txt
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], '*.json'))):
    try:
        d = json.load(open(f))
        it = d['solver']['iterations_per_step']
        print(f"{os.path.basename(f):24s} {d['value']:9.1f} Mcell-upd/s  {d['ms_per_step']:7.3f} ms/step  windows {d['windows']['ms_per_step']}  sweeps {min(i['sweeps'] for i in it)}-{max(i['sweeps'] for i in it)}  pass {d['roofline']['avg_launch_us']} us  {d['config']['flow_field']}")
    except Exception as ex:
        print(os.path.basename(f), 'unreadable:', ex)
PY
