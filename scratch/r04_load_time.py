import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
for (nx, ny, T, K) in [(109, 28, 912, 12), (200, 50, 912, 12), (200, 50, 96, 1)]:
    mesh = cw.synthetic.make_mesh(nx, ny, T, seed=20100529, n_merge=nx, dx=75.0, dy=75.0, depth=3.0, dt=3600.0, velocity=0.3, breathing=0.0, diffusion_coefficient=0.1, period_steps=24)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=86400.0)
    for sweeps in ('0', '128'):
        os.environ['CWR_BOUND_SWEEPS'] = sweeps
        t0 = time.perf_counter()
        model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
        el = time.perf_counter() - t0
        F = model.engine.error_factors()
        print(f'n={mesh["nreal"] + 1} T={T} K={K} CWR_BOUND_SWEEPS={sweeps}: construction {el:.2f} s, factor min/median/max {F[:-1].min():.1f}/{np.median(F[:-1]):.1f}/{F[:-1].max():.1f}, ||J||inf max {model.engine.jacobi_norms().max():.4f}', flush=True)
        model.engine.close()
