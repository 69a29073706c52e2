"""Round 3: sweeps and ms per step of the ping-pong passes (round 2) and the chained passes, from CFL 2.5 to the stiff regime.
usage: r03_stiff.py <K> <steps> <dt> <mode:pingpong|chains> <reps> [more (mode reps) pairs ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport

K, steps, dt = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
pairs = [(sys.argv[i], sys.argv[i + 1]) for i in range(4, len(sys.argv) - 1, 2)]
warm = 5            # (round 5: five -- a batch shape taken for the first time builds its graph, ~10 ms at 1 M cells, and with two warm-up steps that fell into the timed ones)
mesh = cw.synthetic.bench_mesh(warm + steps + 1, dt=dt)
inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=4)
first = None
for mode, reps in pairs:
    os.environ.pop('CWR_NO_CHAINS', None); os.environ.pop('CWR_LOCAL_REPS', None)
    if mode == 'pingpong':
        os.environ['CWR_NO_CHAINS'] = '1'
    if reps != 'auto':
        os.environ['CWR_LOCAL_REPS'] = reps
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    eng = pt.engine
    rho = eng.jacobi_norms()[warm]
    for t in range(warm):
        pt.step(t, tol=1e-12, max_iter=200000, mass_flux=True)
    eng.synchronize()
    sweeps = []
    t0 = time.perf_counter()
    for t in range(warm, warm + steps):
        r = pt.step(t, tol=1e-12, max_iter=200000, mass_flux=True)
        sweeps.append(r.sweeps)
    eng.synchronize()
    el = (time.perf_counter() - t0) / steps
    st = pt.gather_state()
    if first is None:
        first = st
    err = float(np.max(np.abs(st - first)) / np.max(np.abs(first)))
    print(f'1M x {K}  dt={dt:g} s  ||J||inf={rho:.4f}  {mode:8s} reps={reps:4s}: {el * 1e3:8.3f} ms/step  sweeps {min(sweeps)}-{max(sweeps)}  '
          f'flags {r.flags}  max diff vs first config {err:.1e}  resid {r.max_rel_residual:.1e}', flush=True)
    eng.close()
