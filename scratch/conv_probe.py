import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
for nx, K in [(100,1),(100,16),(300,1),(300,16),(1000,1),(1000,16)]:
    mesh = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=40.0, diffusion_coefficient=0.5)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K)
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    for tol in (1e-6, 1e-9, 1e-12):
        pt.engine.set_state(inputs3[0,:mesh['nreal']+1,:])
        try:
            t0=time.time(); r = pt.step(0, tol=tol, max_iter=300, mass_flux=False); el=time.time()-t0
            print(nx, K, tol, 'sweeps', r.sweeps, 'iters', r.iterations, 'restarts', r.restarts, 'resid', r.max_rel_residual, 'ms', el*1e3, flush=True)
        except Exception as e:
            print(nx, K, tol, 'FAIL', e, flush=True)
