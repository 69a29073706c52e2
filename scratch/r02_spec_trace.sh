#!/bin/bash
export TMPDIR=/tmp

rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --constituents 16 > /dev/null 2>&1
f=$(find /tmp/tr -name '*kernel_trace.csv' | head -1)
python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find last k_rhs and print the timeline of ~45 kernels after it: name, stream/queue, start offset, duration
idx=[i for i,r in enumerate(rows) if 'k_rhs' in r['Kernel_Name']]
i0=idx[len(idx)//2]
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0-12:i0+6]:
    print(f"{r['Kernel_Name'][:40]:40s} q{r.get('Queue_Id','?'):>3s} start {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}")
PY
