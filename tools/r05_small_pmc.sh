#!/bin/bash
# LDS counters of k_small_jacobi at 2 943 cells (counter-only passes, no trace domains beside the kernel trace)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05zj}
export TMPDIR=/tmp
out=gpurun_out/${tag}_small_pmc.txt; : > "$out"
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  d=gpurun_out/${tag}_pmc
  rm -rf "$d"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$d" -o p -- python3 tools/small_step_profile.py --K 1 --steps 30 --warmup 5 --label pmc > "$d.log" 2>&1 || { echo "FAILED: $set" | tee -a "$out"; tail -3 "$d.log" | tee -a "$out"; continue; }
  f=$(find "$d" -name '*counter_collection.csv' | head -1)
  python3 - "$f" >> "$out" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_small_jacobi' in r['Kernel_Name']:
        a = acc[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (c, v) in acc.items():
    print(f'k_small_jacobi  {k:24s} launches {c:4d}  per launch {v / c:14.1f}')
PY
done
cat "$out"
