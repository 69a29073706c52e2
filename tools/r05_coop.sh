#!/bin/bash
# Round 5: the one-launch solver with several parts per constituent (k_small_jacobi<RPT, true>) at "~10 k cells": A/B against the
# multi-launch passes (CWR_SMALL_MAX_CELLS=0), parts / depth sweeps, kernel budget.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05zr}
out=gpurun_out/${tag}_coop.txt; : > "$out"
export TMPDIR=/tmp
run() { timeout -k 10 120 python3 tools/small_step_profile.py "$@" 2>&1 | grep -E "SMALLSTEP|rror|cwr:" | tee -a "$out"; }
for cfg in "160 50 1" "160 50 12" "200 50 1" "200 50 12" "300 60 1" "300 60 12"; do set -- $cfg
  run --nx $1 --ny $2 --merge 0 --K $3 --label "parts auto, depth 4"
  CWR_SMALL_MAX_CELLS=0 run --nx $1 --ny $2 --merge 0 --K $3 --label "multi-launch passes"
done
for d in 2 3 6 8; do CWR_SMALL_DEPTH=$d run --nx 200 --ny 50 --merge 0 --K 12 --label "depth $d"; done
for p in 5 6 8; do CWR_SMALL_PARTS=$p run --nx 200 --ny 50 --merge 0 --K 12 --label "parts $p"; done
for p in 5 6 8; do CWR_SMALL_PARTS=$p run --nx 200 --ny 50 --merge 0 --K 1 --label "parts $p"; done
d=gpurun_out/${tag}_trace
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o t -- python3 tools/small_step_profile.py --nx 200 --ny 50 --merge 0 --K 12 --steps 60 --label rocprofv3 > "$d.log" 2>&1
grep SMALLSTEP "$d.log" | tee -a "$out"
tr=$(find "$d" -name '*kernel_trace.csv' | head -1)
python3 tools/trace_budget.py "$tr" --steps 50 --label "10 000 cells x 12, several parts" >> "$out" 2>&1
rm -rf "$d"
cat "$out"
