#!/bin/bash
# Functional rehearsal of `python bench.py --gpus N` on ONE GPU through the stream-asynchronous RCCL stand-in (tests/mock_rccl): the ranks
# share the card, so times mean little; the per-rank report is what is shown.  usage: bash tools/r06_rehearse.sh [N ...]
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
python3 -c "import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle'); from test_gpu_multirank import build_mock; print(build_mock())" > /tmp/mock_path.txt 2>/dev/null || exit 1
mock=$(tail -1 /tmp/mock_path.txt)
out=gpurun_out/r06_multirank_rehearsal.txt
: > "$out"
for N in "${@:-2 4 6}"; do
  for n in $N; do
    line=$(CWR_RCCL_LIB=$mock CWR_BENCH_DEVICE=0 CWR_MOCK_ASYNC=2 timeout -k 10 500 python bench.py --gpus $n --steps 6 --warmup 3 --windows 2 $REHEARSE_ARGS 2>gpurun_out/r06_rehearse_$n.err | tail -1)
    python3 - "$n" "$line" <<'PY' | tee -a "$out"
import json, sys
n, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
except Exception as ex:
    print(f'--gpus {n}: FAILED ({ex}) {line[:300]}'); sys.exit(0)
it = d['solver']['iterations_per_step']
print(f"--gpus {n}: {d['config']['partition']}, chained {d['solver']['chained_passes']}, {d['ms_per_step']} ms/step on the shared GPU, sweeps {[i['sweeps'] for i in it]}, "
      f"exchanges {[i['exchanges'] for i in it]}, overlapped {[i['overlapped'] for i in it]}, checks {[i['checks'] for i in it]}")
for r in d.get('ranks', []):
    print(f"   rank {r['rank']} (range {r.get('cell_range')}, group {r.get('constituent_group')}, {r.get('constituents')} constituents): rows computed / owned {r['computed_rows']} / {r['owned_rows']} (halo {r['halo_rows']}, {r['peers']} peers, depth {r['halo_depth']}), tiles {r['tiles']} x {r['tile_rows']} on {r['grid']} blocks; "
          f"passes {r['passes_per_step']} x {r['mean_pass_us']} us per step; exchanges alone {r['exchanges_alone']} ({r['exchanges_alone_us']} us), beside compute {r['exchanges_beside_compute']} ({r['exchanges_beside_compute_us']} us), "
          f"all-reduces {r['allreduces']} ({r['allreduces_us']} us), checks {r['checks']} (host inside them {r['check_host_wait_us']} us) over {r['steps_profiled']} steps; "
          f"stand-alone step {r.get('standalone_ms_per_step')} ms at {r.get('standalone_sweeps')} sweeps {r.get('standalone_error', '')}")
if 'compute_side_ceiling' in d:
    print(f"   compute-side ceiling: {d['compute_side_ceiling']}")
PY
  done
done
