#!/bin/bash
# where does a step of a small engine go: kernel time vs wall (rocprofv3 kernel trace of scratch/r04_small.py)
export TMPDIR=/tmp
for cs in "band200x50 12" "sq354 16"; do
  tag=$(echo $cs | tr ' ' '_'); rm -rf /tmp/tr_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$tag -o run -- python3 scratch/r04_small.py $cs default= > gpurun_out/r04j_trace_$tag.txt 2>/tmp/tr_$tag.err
  f=$(find /tmp/tr_$tag -name '*kernel_stats.csv' | head -1)
  echo "== $cs"; cat gpurun_out/r04j_trace_$tag.txt | tail -1
  python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time total ms', tot/1e6)
for r in rows[:8]: print(f"  {r['Name'][:50]:50s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
  t=$(find /tmp/tr_$tag -name '*kernel_trace.csv' | head -1)
  python - $t <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
# steps: 20 total (4 warm + 16 timed); look at the gaps between consecutive k_sq_tiled launches
ts=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in rows]
ts.sort()
gaps=[]; durs=[]
for a,b in zip(ts[:-1],ts[1:]):
    if 'k_sq_tiled' in a[2] and 'k_sq_tiled' in b[2]:
        gaps.append(b[0]-a[1]); durs.append(a[1]-a[0])
import statistics
print('k_sq_tiled -> k_sq_tiled: n', len(gaps), 'median gap us', statistics.median(gaps)/1e3, 'median duration us', statistics.median(durs)/1e3)
PY
done
