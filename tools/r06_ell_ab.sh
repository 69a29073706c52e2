#!/bin/bash
# Round 6 A/B of the dominant pass: wave-sliced entry layout (CWR_TCL_ELL=1, default) against CSR order (CWR_TCL_ELL=0), same box, alternating.
# usage: tools/r06_ell_ab.sh [K ...]
mkdir -p gpurun_out
out=gpurun_out/r06_ell_ab.txt
run() {  # K label env...
  K=$1; label=$2; shift; shift
  line=$(env "$@" python bench.py --constituents $K --steps 20 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | tail -1)
  python - "K=$K $label" "$line" <<'PY' | tee -a $out
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    it = d['solver']['iterations_per_step']
    sw = [i['sweeps'] for i in it]
    r = d['roofline'] or {}
    print(f"{label:34s} {d['ms_per_step']:.3f} ms/step (windows {min(d['windows']['ms_per_step']):.3f}-{max(d['windows']['ms_per_step']):.3f}), pass {r.get('avg_launch_us')} us x {r.get('launches_timed', 0) / d['steps']:.1f} per step, "
          f"sweeps {min(sw)}-{max(sw)}, reps {d['solver']['tile_local_applications']}, resid {d['solver']['max_rel_residual']:.1e}, {d['value']:.0f} Mcell-updates/s")
except Exception as ex:
    print(f'{label}: FAILED {ex} {line[:200]}')
PY
}
echo "# $(date -u +%H:%M:%S)" | tee -a $out
for K in "${@:-16 1}"; do for k in $K; do
  run $k "CSR order" CWR_TCL_ELL=0
  run $k "sliced" CWR_TCL_ELL=1
  run $k "CSR order (again)" CWR_TCL_ELL=0
  run $k "sliced (again)" CWR_TCL_ELL=1
  run $k "sliced, 3 applications" CWR_TCL_ELL=1 CWR_LOCAL_REPS=3
done; done
