#!/bin/bash
# parameter sweep of the several-parts one-launch solver: depth x parts at 8 k / 10 k / 14 k cells, K = 1 and 12
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05zs}
out=gpurun_out/${tag}_coop_sweep.txt; : > "$out"
run() { timeout -k 10 120 python3 tools/small_step_profile.py "$@" 2>&1 | grep -E "SMALLSTEP|rror|cwr:" | tee -a "$out"; }
for cfg in "160 50" "200 50" "280 50" "120 110"; do set -- $cfg
  for K in 1 12; do
    CWR_SMALL_MAX_CELLS=0 run --nx $1 --ny $2 --merge 0 --K $K --steps 100 --label "multi-launch passes"
    for d in 6 8 12; do
      for p in 0 6 8; do
        CWR_SMALL_DEPTH=$d CWR_SMALL_PARTS=$p run --nx $1 --ny $2 --merge 0 --K $K --steps 100 --label "depth $d parts $p"
      done
    done
  done
done
