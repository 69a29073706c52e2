#!/bin/bash
# Round 6 A/B of the dominant pass: the SAME tiled kernel over J's own pattern (CWR_TCL_POWER=1: a pass = `reps` plain Jacobi sweeps per tile,
# ring-1 columns, constant bhat, no numeric J^2 and no c2 sweep per step) against the J^2 pattern (default), same box, alternating.
# usage: tools/r06_power_ab.sh [K=16] [extra bench args]      -> one line per variant: ms/step, pass us, sweeps, launches
K=${1:-16}; shift
out=gpurun_out/r06_power_ab_K$K.txt
mkdir -p gpurun_out
run() {  # label, env...
  label=$1; shift
  line=$(env "$@" python bench.py --constituents $K --steps 20 --warmup 5 --no-cpu-baseline --no-pmc "${EXTRA[@]}" 2>/dev/null | tail -1)
  python - "$label" "$line" <<'PY' | tee -a $out
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    it = d['solver']['iterations_per_step']
    sw = [i['sweeps'] for i in it]
    r = d['roofline'] or {}
    print(f"{label:34s} {d['ms_per_step']:.3f} ms/step (windows {min(d['windows']['ms_per_step']):.3f}-{max(d['windows']['ms_per_step']):.3f}), pass {r.get('avg_launch_us')} us x {r.get('launches_timed', 0) / d['steps']:.1f} per step, "
          f"sweeps {min(sw)}-{max(sw)}, bicgstab {max(i['bicgstab'] for i in it)}, reps {d['solver']['tile_local_applications']}, resid {d['solver']['max_rel_residual']:.1e}")
except Exception as ex:
    print(f'{label}: FAILED {ex} {line[:200]}')
PY
}
EXTRA=("$@")
echo "# K=$K $(date -u +%H:%M:%S) extra: ${EXTRA[*]}" | tee -a $out
run "J^2 default" CWR_X=0
for reps in 3 4 5 6; do run "J power 1, reps $reps" CWR_TCL_POWER=1 CWR_LOCAL_REPS=$reps; done
run "J^2 default (again)" CWR_X=0
run "J^2 reps 3" CWR_LOCAL_REPS=3
