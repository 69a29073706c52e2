#!/bin/bash
# Round 5: the reference's own mesh sizes (2 943 cells): parity, A/B of the copy-free read-out of k_small_jacobi, kernel-level budget.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05zf}
out=gpurun_out/${tag}_small_mesh.txt
: > "$out"
export TMPDIR=/tmp
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_integration_doc.py tests/test_gpu_exit.py -x -q -m gpu > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a "$out"; tail -3 gpurun_out/${tag}_pytest.log | tee -a "$out"
for K in 1 12; do
  for rep in 1 2; do
    timeout -k 10 120 python3 tools/small_step_profile.py --K $K --label "default" 2>&1 | grep SMALLSTEP | tee -a "$out"
    CWR_NO_NOTE=1 timeout -k 10 120 python3 tools/small_step_profile.py --K $K --label "CWR_NO_NOTE=1" 2>&1 | grep SMALLSTEP | tee -a "$out"
  done
  timeout -k 10 120 python3 tools/small_step_profile.py --K $K --no-flux --label "default" 2>&1 | grep SMALLSTEP | tee -a "$out"
done
d=gpurun_out/${tag}_trace
for K in 1 12; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d$K" -o t -- python3 tools/small_step_profile.py --K $K --steps 60 --label rocprofv3 > "$d$K.log" 2>&1
  grep SMALLSTEP "$d$K.log" | tee -a "$out"
  tr=$(find "$d$K" -name '*kernel_trace.csv' | head -1)
  python3 tools/trace_budget.py "$tr" --steps 50 --label "2943 cells x $K" >> "$out" 2>&1
  rm -rf "$d$K"
done

for sz in "30 12" "50 38" "90 45"; do set -- $sz
  timeout -k 10 120 python3 tools/small_step_profile.py --nx $1 --ny $2 --merge 40 --K 1 --label "default" 2>&1 | grep SMALLSTEP | tee -a "$out"
done
cat "$out"
