"""Round 3: chained in-place passes against today's ping-pong passes.  One engine per configuration, same mesh and inputs.
usage: r03_chain.py <mesh: bench|quadNX> <K> <dt> <steps> [reps list]   (prints one line per configuration)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
from clearwater_riverine_amd import schedule as sch

which, K, dt, steps = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
reps_list = [int(v) for v in sys.argv[5:]] or [2, 3, 4]
warm = 2
if which == 'bench':
    mesh = cw.synthetic.bench_mesh(warm + steps + 1, dt=dt)
else:
    nx = int(which[4:])
    mesh = cw.synthetic.make_mesh(nx, nx, warm + steps + 1, seed=4, dt=dt, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=4)
n = mesh['nreal'] + 1
ref_state = None
for mode in ['pingpong'] + [f'chain{s}' for s in (2,)]:
    for reps in reps_list:
        os.environ['CWR_LOCAL_REPS'] = str(reps)
        pt = PartitionedTransport(mesh, inputs3, 0, 1)
        eng = pt.engine
        ok, ntiles, grid, TR = eng.tiling_info()
        t_build = 0.0
        if mode.startswith('chain'):
            t0 = time.perf_counter()
            lm = pt.local
            sc = sch.chain_schedule(lm.face1, lm.face2, np.asarray(mesh['face_flow'][warm])[lm.edge_global], lm.n_rows, TR, ntiles, grid,
                                    streams_per_block=int(mode[5:]))
            t_build = time.perf_counter() - t0
            eng.set_tile_schedule(sc)
        sweeps = []
        for t in range(warm):
            pt.step(t, tol=1e-12, max_iter=200000, mass_flux=True)
        eng.synchronize()
        t0 = time.perf_counter()
        for t in range(warm, warm + steps):
            r = pt.step(t, tol=1e-12, max_iter=200000, mass_flux=True)
            sweeps.append(r.sweeps)
        eng.synchronize()
        el = (time.perf_counter() - t0) / steps
        st = pt.owned_state()
        if ref_state is None:
            ref_state = st
        err = float(np.max(np.abs(st - ref_state)) / np.max(np.abs(ref_state)))
        print(f'{which} K={K} dt={dt:g} {mode} reps={reps}: {el * 1e3:.3f} ms/step, sweeps {sweeps[:6]}..{sweeps[-1]}, kernel {r.sweep_kernel}, '
              f'tiles {ntiles} grid {grid}, schedule build {t_build * 1e3:.0f} ms, max diff vs first {err:.1e}, resid {r.max_rel_residual:.1e}', flush=True)
        eng.close()
