"""Round 4: engines below the chain threshold (one rank of 8 of the 1 M-cell mesh = 125 k cells; 60 k; the Ohio-sized band).
One engine per environment combination, same mesh and inputs; prints ms per step and sweeps.
usage: r04_small.py <case: sq354|sq245|band200x50|bend1026x256@1.0|...> <K> [label=ENV1=v,ENV2=v ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport

case, K = sys.argv[1], int(sys.argv[2])
combos = sys.argv[3:] or ['default=']
steps, warm = 16, 4
if case.startswith('bend'):
    # bendNXxNY[@theta0]: the jittered, 5 %-merged channel of make_mesh laid along a meander (synthetic.bend_channel)
    dims, _, th = case[4:].partition('@')
    nx, ny = (int(v) for v in dims.split('x'))
    mesh = cw.synthetic.make_mesh(nx, ny, warm + steps + 1, seed=4, dt=float(os.environ.get('MID_DT', '40')), diffusion_coefficient=0.5, n_merge=int(0.05 * nx * ny))
    if th and float(th) > 0:
        mesh = cw.synthetic.bend_channel(mesh, float(th), wavelength=float(os.environ.get('BEND_WAVELENGTH', '0')) or None)
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=4)
elif case.startswith('sq'):
    nx = int(case[2:])
    mesh = cw.synthetic.make_mesh(nx, nx, warm + steps + 1, seed=4, dt=float(os.environ.get('MID_DT', '40')), diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=4)
else:
    nx, ny = (int(v) for v in case[4:].split('x'))
    mesh = cw.synthetic.make_mesh(nx, ny, warm + steps + 1, seed=20100529, n_merge=0, dx=75.0, dy=75.0, depth=3.0, dt=3600.0,
                                  velocity=0.3, breathing=0.0, diffusion_coefficient=0.1, period_steps=24)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=86400.0)
ref = None
touched = set()
for combo in combos:
    label, _, envs = combo.partition('=')
    for k in touched: os.environ.pop(k, None)
    touched = set()
    for kv in filter(None, envs.split(',')):
        k, _, v = kv.partition(':')
        os.environ[k] = v; touched.add(k)
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    for t in range(warm): pt.step(t, tol=1e-12)
    pt.engine.synchronize(); t0 = time.perf_counter(); sw = []
    for t in range(warm, warm + steps): sw.append(pt.step(t, tol=1e-12).sweeps)
    pt.engine.synchronize(); el = (time.perf_counter() - t0) / steps
    ok, ntiles, grid, TR = pt.engine.tiling_info()
    sched = pt.engine.get_tile_schedule()[0]
    st = pt.gather_state()
    if ref is None: ref = st
    err = float(np.max(np.abs(st - ref)) / np.max(np.abs(ref)))
    print(f'{case} n={mesh["nreal"] + 1} K={K} [{label}] {pt.numbering}: tiles {ntiles} x {TR} rows, grid {grid} ({ntiles / max(grid, 1):.1f}/block), chains {"on " if sched is not None else "off"}: '
          f'{el * 1e3:.3f} ms/step, sweeps {min(sw)}-{max(sw)}, diff vs first {err:.1e}', flush=True)
    pt.engine.close()
