#!/usr/bin/env python3
"""Where a facade update() spends its time at the reference's own mesh size (2 943 cells): cProfile over 2 000 updates.
usage: facade_profile.py [K] [--plain]   (--plain: wall time only, no profiler)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw

K = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
steps = 2000
mesh = cw.synthetic.make_mesh(109, 28, steps + 2, seed=20100529, n_merge=109, dx=75.0, dy=75.0, depth=3.0, dt=3600.0, velocity=0.3, breathing=0.0,
                              diffusion_coefficient=0.1, period_steps=24)
inputs3 = cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=86400.0)
names = [f'c{k}' for k in range(K)]
model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)})
for _ in range(50):
    model.update()
def run():
    for _ in range(steps - 50):
        model.update()
t0 = time.perf_counter()
if '--plain' in sys.argv:
    run()
else:
    pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
el = time.perf_counter() - t0
print(f'FACADE n={mesh["nreal"] + 1} K={K}: {el / (steps - 50) * 1e3:.4f} ms per update() ({"plain" if "--plain" in sys.argv else "under cProfile"})')
if '--plain' not in sys.argv:
    pstats.Stats(pr).sort_stats('tottime').print_stats(18)
