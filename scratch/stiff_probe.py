"""Stiff steps: block-asynchronous Jacobi passes vs BiCGSTAB as a function of the time step (CFL)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
nx, ny, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for dt in [float(v) for v in sys.argv[4:]]:
    mesh = cw.synthetic.make_mesh(nx, ny, 4, seed=3, dt=dt, breathing=0.0, n_merge=10)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=86400.0)
    for solver in ('jacobi', 'bicgstab'):
        pt = PartitionedTransport(mesh, inputs3, 0, 1)
        try:
            pt.step(0, tol=1e-12, max_iter=200000, solver=solver, mass_flux=False)
            t0 = time.perf_counter(); r = pt.step(1, tol=1e-12, max_iter=200000, solver=solver, mass_flux=False); pt.engine.synchronize(); el = time.perf_counter() - t0
            print(f'{nx}x{ny} K={K} dt={dt:g} {solver}: sweeps {r.sweeps} bicgstab {r.iterations} restarts {r.restarts} {el*1e3:.2f} ms resid {r.max_rel_residual:.1e}', flush=True)
        except Exception as exc:
            print(f'{nx}x{ny} K={K} dt={dt:g} {solver}: FAILED {exc}', flush=True)
