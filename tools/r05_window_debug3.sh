export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --windows 1"
show() { python -c "
import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], d['ms_per_step'], d['windows']['ms_per_step'], 'pass', d['roofline']['avg_launch_us'], 'sweeps', sorted(set(i['sweeps'] for i in d['solver']['iterations_per_step'])))" $1 "$2"; }
$B > /dev/null 2>&1
for K in 16 1; do
  $B --constituents $K > /tmp/a.json 2>/dev/null; show /tmp/a.json "K=$K resident"
  $B --constituents $K --flow-window 8 > /tmp/a.json 2>/dev/null; show /tmp/a.json "K=$K window 8"
  CWR_WINDOW_EAGER=1 $B --constituents $K --flow-window 8 > /tmp/a.json 2>/dev/null; show /tmp/a.json "K=$K window 8, loads enqueued at the call (eager)"
  CWR_WINDOW_DEBUG=1 $B --constituents $K --flow-window 8 > /tmp/a.json 2>/dev/null; show /tmp/a.json "K=$K window 8, bookkeeping only"
  $B --constituents $K > /tmp/a.json 2>/dev/null; show /tmp/a.json "K=$K resident again"
done
