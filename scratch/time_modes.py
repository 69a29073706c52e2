import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
for K in (16, 1):
    mesh = cw.synthetic.make_mesh(1000, 1000, 4, seed=4, dt=40.0, diffusion_coefficient=0.5)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K)
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    pt.step(0, tol=1e-12, mass_flux=False)
    for rep in range(2):
        pt.engine.profile_read()
        t0 = time.perf_counter(); r = pt.step(1 + rep, tol=1e-12, mass_flux=False, profile=True); el = time.perf_counter() - t0
        n, us = pt.engine.profile_read()
        print(f'K={K} step: {el*1e3:.2f} ms, sweeps {r.sweeps}, launches {n}, avg operator launch {us/max(n,1):.1f} us, back-to-back {pt.engine.time_apply(1, reps=40):.1f} us', flush=True)
