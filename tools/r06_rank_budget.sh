#!/bin/bash
# price "constituent groups x cell ranges" on 8 GPUs for the 1 M-cell x 16 workload -- the K systems share A and never talk
# to each other (transport.py:231-249), so 8 GPUs may be R cell ranges x G groups of 16 / G constituents.  One rank of each arrangement stepped
# ALONE (tools/rank_step_profile.py: real partition, one-rank communicator, perfect halo data); ranks whose own iteration cannot converge alone
# run the sweep count of the arrangement's rank 0 (which can).   usage: bash tools/r06_rank_budget.sh <tag>
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r06e}
out=gpurun_out/${tag}_rank_budget.txt
: > "$out"
run() {  # mesh K rank world [fixed]
  local fixed=${5:+--fixed-sweeps $5}
  timeout -k 10 400 python3 tools/rank_step_profile.py --mesh "$1" --K "$2" --rank "$3" --world "$4" --steps 10 --warmup 4 $fixed 2>&1 | grep RANKSTEP | tee -a "$out"
}
for arr in "16 8" "8 4" "4 2" "2 1"; do   # K per group, cell ranges: 1 group x 8 ranges, 2 x 4, 4 x 2, 8 x 1
  set -- $arr; K=$1; W=$2
  echo "--- groups of $K constituents x $W cell ranges  $(date +%T)" | tee -a "$out"
  line=$(run 1m $K 0 $W)
  echo "$line" >> /dev/null
  sw=$(echo "$line" | sed -n 's/.*sweeps=[0-9]*-\([0-9]*\).*/\1/p')
  if [ "$W" -gt 1 ]; then
    [ -z "$sw" ] && sw=51
    [ $((sw % 2)) -eq 0 ] && sw=$((sw + 1))
    if [ "$W" -gt 2 ]; then run 1m $K $((W / 2 - 1)) $W $sw; fi
    run 1m $K $((W - 1)) $W $sw
  fi
done
cat "$out"
