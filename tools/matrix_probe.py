#!/usr/bin/env python3
"""A sweep over mesh sizes, time steps and constituent counts, a few dozen steps each: per case the step time (median / max), the
range of sweeps, BiCGSTAB iterations and flags -- to SPOT pathologies (run-away batches, fallbacks, clamps), not to benchmark.
usage: matrix_probe.py [steps=30]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cases = []
for (nx, ny) in [(109, 28), (200, 50), (300, 60), (400, 100), (600, 200)]:
    for dt in (600.0, 3600.0, 14400.0):
        for K in (1, 4, 12, 16):
            cases.append((nx, ny, dt, K))
only = os.environ.get('PROBE_ONLY')            # e.g. "300x60"
for (nx, ny, dt, K) in cases:
    if only and only != f'{nx}x{ny}':
        continue
    mesh = cw.synthetic.make_mesh(nx, ny, steps + 2, seed=20100529, n_merge=nx * ny // 40, dx=75.0, dy=75.0, depth=3.0, dt=dt, velocity=0.3, breathing=0.1,
                                  diffusion_coefficient=0.1, period_steps=24, n_dry=nx * ny // 200)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=86400.0)
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    ms, sw, its, flags, kern = [], [], 0, 0, set()
    try:
        for t in range(steps):
            t0 = time.perf_counter()
            r = pt.engine.step(t, tol=1e-12)
            pt.engine.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
            sw.append(r.sweeps); its = max(its, r.iterations); flags |= r.flags; kern.add(r.sweep_kernel)
        note = ''
        if max(sw[2:]) > 2.0 * np.median(sw[2:]) or its > 0 or flags:
            note = '   <<< LOOK'
        print(f'{nx}x{ny} n={mesh["nreal"] + 1} dt={dt:g} K={K}: median {np.median(ms[2:]):.3f} max {max(ms[2:]):.3f} ms/step, sweeps {min(sw[2:])}-{max(sw[2:])}, bicgstab {its}, '
              f'flags {flags}, kernel {sorted(kern)}{note}', flush=True)
    except Exception as exc:
        print(f'{nx}x{ny} dt={dt:g} K={K}: {type(exc).__name__}: {str(exc)[:160]}   <<< LOOK', flush=True)
    pt.engine.close()
