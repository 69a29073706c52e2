#!/bin/bash
# Round 5: BASELINE configs 1 / 2 at "~10 k cells" (the band of tests/models/ohio_like.py at 8 000 and 10 000 cells): step time and
# kernel-level budget of the multi-launch path that meshes above 4 096 cells take.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05zq}
out=gpurun_out/${tag}_mid_mesh.txt; : > "$out"
export TMPDIR=/tmp
timeout -k 10 200 python3 -m pytest tests/test_gpu_outputs.py -x -q -m gpu -k in_place 2>&1 | tail -2 | tee -a "$out"
for cfg in "160 50 1" "160 50 12" "200 50 1" "200 50 12" "300 60 12"; do set -- $cfg
  timeout -k 10 120 python3 tools/small_step_profile.py --nx $1 --ny $2 --merge 0 --K $3 --label "default" 2>&1 | grep SMALLSTEP | tee -a "$out"
done
for cfg in "200 50 1" "200 50 12"; do set -- $cfg
  d=gpurun_out/${tag}_trace_$3
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o t -- python3 tools/small_step_profile.py --nx $1 --ny $2 --merge 0 --K $3 --steps 60 --label rocprofv3 > "$d.log" 2>&1
  grep SMALLSTEP "$d.log" | tee -a "$out"
  tr=$(find "$d" -name '*kernel_trace.csv' | head -1)
  python3 tools/trace_budget.py "$tr" --steps 50 --label "10 000 cells x $3" >> "$out" 2>&1
  rm -rf "$d"
done
cat "$out"
