#!/bin/bash
# VERDICT r04 task 1a: kernel-level budget of one rank of 8 stepped alone (configs 4 / 5 and the bench workload), next to the
# single-GPU step of the same mesh.  Usage (on the GPU box): bash tools/r05_rank_budget.sh <tag> [cases...]
#   case = mesh:K:rank:world[:fixed sweeps]   (default list below)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05a}; shift
cases=("$@")
[ ${#cases[@]} -eq 0 ] && cases=(1m:16:0:8 1m:16:3:8 1m:16:7:8 1m:16:0:1 1m:1:0:8 1m:1:3:8 1m:1:7:8 1m:1:0:1 4m:16:0:8 4m:16:3:8 4m:16:7:8)
out=gpurun_out/${tag}_rank_budget.txt
mkdir -p gpurun_out/${tag}_traces
: > "$out"
export TMPDIR=/tmp
for c in "${cases[@]}"; do
  IFS=: read -r mesh K rank world fixed <<< "$c"
  EXTRA_ARGS="${fixed:+--fixed-sweeps $fixed}"
  name=${mesh}_K${K}_r${rank}of${world}
  d=gpurun_out/${tag}_traces/$name
  echo "--- $name $(date +%T)"
  # (a) plain run: the step time without the profiler's per-launch overhead
  timeout -k 10 560 python3 tools/rank_step_profile.py --mesh "$mesh" --K "$K" --rank "$rank" --world "$world" --steps 10 --warmup 4 ${EXTRA_ARGS} \
      > "$d.plain.log" 2>&1 || { echo "FAILED (plain) $name rc=$?" | tee -a "$out"; tail -5 "$d.plain.log"; continue; }
  grep RANKSTEP "$d.plain.log" | sed 's/^RANKSTEP/RANKSTEP[plain]/' | tee -a "$out"
  # (b) the same under rocprofv3: the kernel-level budget
  timeout -k 10 560 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o t -- \
      python3 tools/rank_step_profile.py --mesh "$mesh" --K "$K" --rank "$rank" --world "$world" --steps 10 --warmup 4 ${EXTRA_ARGS} \
      > "$d.log" 2>&1 || { echo "FAILED $name rc=$?" | tee -a "$out"; tail -5 "$d.log"; continue; }
  grep RANKSTEP "$d.log" | sed 's/^RANKSTEP/RANKSTEP[rocprofv3]/' | tee -a "$out"
  tr=$(find "$d" -name '*kernel_trace.csv' | head -1)
  python3 tools/trace_budget.py "$tr" --steps 10 --label "$name" >> "$out" 2>&1
  st=$(find "$d" -name '*kernel_stats.csv' | head -1)
  [ -n "$st" ] && cp "$st" gpurun_out/${tag}_traces/${name}_kernel_stats.csv
  rm -rf "$d"          # (the raw trace is large; the stats csv and the budget stay)
done
cat "$out"
