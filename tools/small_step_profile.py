#!/usr/bin/env python3
"""Engine-only step time at the reference's own mesh sizes (the Ohio-sized band of tests/models/ohio_like.py: 109 x 28 = 2 943 cells,
dt = 3600 s, CFL ~ 18), for a kernel-level budget under rocprofv3 (tools/trace_budget.py) and for A/B runs of host-path knobs.
usage: small_step_profile.py [--nx 109 --ny 28 --merge 109] [--K 1] [--steps 200] [--warmup 20] [--no-flux]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw

ap = argparse.ArgumentParser()
ap.add_argument('--nx', type=int, default=109); ap.add_argument('--ny', type=int, default=28); ap.add_argument('--merge', type=int, default=109)
ap.add_argument('--K', type=int, default=1); ap.add_argument('--steps', type=int, default=200); ap.add_argument('--warmup', type=int, default=20)
ap.add_argument('--no-flux', action='store_true'); ap.add_argument('--label', default='')
a = ap.parse_args()
T = 48
mesh = cw.synthetic.make_mesh(a.nx, a.ny, T, seed=20100529, n_merge=a.merge, dx=75.0, dy=75.0, depth=3.0, dt=3600.0, velocity=0.3, breathing=0.0,
                              diffusion_coefficient=0.1, period_steps=24)
inputs3 = cw.synthetic.boundary_input_array(mesh, a.K, inlet_period_s=86400.0)
n = mesh['nreal'] + 1
from clearwater_riverine_amd.distributed import PartitionedTransport
pt = PartitionedTransport(mesh, inputs3, 0, 1)      # (derives dt and the face distances as the facade does; one engine, no comm)
eng = pt.engine
for s in range(a.warmup):
    eng.step(s % T, mass_flux=not a.no_flux)
eng.synchronize()
t0 = time.perf_counter(); sw = []
for s in range(a.steps):
    sw.append(eng.step((a.warmup + s) % T, mass_flux=not a.no_flux).sweeps)
eng.synchronize()
el = (time.perf_counter() - t0) / a.steps
print(f'SMALLSTEP {a.label} n={n} K={a.K} flux={not a.no_flux}: {el * 1e3:.4f} ms/step over {a.steps} steps, sweeps {min(sw)}-{max(sw)}', flush=True)
eng.close()
