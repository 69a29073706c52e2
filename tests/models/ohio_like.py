import sys, os, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'oracle'))
import numpy as np
import clearwater_riverine_amd as cw
import cwr_oracle as oracle
for (nx, ny, nm, K) in [(109, 28, 109, 1), (109, 28, 109, 12), (160, 50, 0, 1), (160, 50, 0, 12), (200, 50, 0, 12)]:
    steps = 40
    mesh = cw.synthetic.make_mesh(nx, ny, steps, seed=20100529, n_merge=nm, dx=75.0, dy=75.0, depth=3.0, dt=3600.0,
                                  velocity=0.3, breathing=0.0, diffusion_coefficient=0.1, period_steps=24)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=86400.0)
    n = mesh['nreal'] + 1
    names = [f'c{k}' for k in range(K)]
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm_: inputs3[:, :, k].copy() for k, nm_ in enumerate(names)})
    model.update()
    t0 = time.perf_counter()
    its = []
    for s in range(1, steps):
        model.update(); its.append((model.last_step.sweeps, model.last_step.iterations))
    gpu = (time.perf_counter() - t0) / (steps - 1)
    # engine-only (no per-step D2H of state and fluxes)
    eng = model.engine
    eng.set_state(inputs3[0, :n, :])
    t0 = time.perf_counter()
    for s in range(steps):
        eng.step(s, mass_flux=True)
    eng_only = (time.perf_counter() - t0) / steps
    oracle.derive_coefficients(mesh)
    ref = oracle.OracleModel(mesh, {nm_: inputs3[:, :, k].copy() for k, nm_ in enumerate(names)})
    t0 = time.perf_counter()
    for s in range(steps):
        ref.update()
    cpu = (time.perf_counter() - t0) / steps
    err = max(np.nanmax(np.abs(model.mesh[nm_][steps] - ref.constituent_dict[nm_].state[steps])) / np.nanmax(np.abs(ref.constituent_dict[nm_].state[steps])) for nm_ in names)
    a = np.abs(mesh['advection_coeff'][0]).astype(float); out = np.zeros(n); np.add.at(out, mesh['edges_face1'], np.maximum(mesh['advection_coeff'][0], 0))
    print(f'n={n} K={K}: facade {gpu*1e3:.2f} ms/step, engine {eng_only*1e3:.2f} ms/step, oracle CPU {cpu*1e3:.2f} ms/step, '
          f'CFL~{(out*3600/mesh["volume"][0,:n]).max():.0f}, iters (sweeps,bicg) last {its[-1]}, rel err {err:.2e}', flush=True)
