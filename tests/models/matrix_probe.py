#!/usr/bin/env python3
"""A sweep over river-band meshes (75 m cells, 2.5 % merged, 0.5 % dry cells, breathing volumes), time steps and constituent counts, a
few dozen steps each.  Per case: the step time (median / max), the range of sweeps, BiCGSTAB iterations, flags, the range of the
a-posteriori error factor F over the levels -- and, since round 6 (VERDICT r05 "next" 1a), for meshes the oracle finishes in seconds
(<= 39 000 cells), the comparison with the oracle's SuperLU result after EVERY step: the max-norm error relative to the peak and
the worst element-wise ratio |a - b| / (1e-6 |b| + 1e-12 max|b|) -- the bar of tests/util.rel_err; <= 1 passes.

The oracle is K-independent here (constituent k of synthetic.boundary_input_array is (k + 1) x constituent 0, and the system is
linear): one oracle run per (mesh, dt), in a process pool started BEFORE anything touches the GPU.

usage: matrix_probe.py [steps=30]          PROBE_ONLY=200x50  PROBE_DT=3600  PROBE_K=1,12  PROBE_ORACLE_MAX=39000  PROBE_WORKERS=12
(lives under tests/: nothing under tools/ or the package imports oracle/)"""
import os, sys, time, warnings
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'oracle')); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SIZES = [(109, 28), (200, 50), (300, 60), (400, 100), (600, 200)]
DTS = [float(v) for v in os.environ.get('PROBE_DT', '600,3600,14400').split(',')]
KS = [int(v) for v in os.environ.get('PROBE_K', '1,4,12,16').split(',')]
ORACLE_MAX = int(os.environ.get('PROBE_ORACLE_MAX', '39000'))
only = os.environ.get('PROBE_ONLY')            # e.g. "300x60"
if only:
    SIZES = [s for s in SIZES if f'{s[0]}x{s[1]}' in only.split(',')]


def band(nx, ny, dt):
    import clearwater_riverine_amd as cw
    return cw.synthetic.make_mesh(nx, ny, steps + 2, seed=20100529, n_merge=nx * ny // 40, dx=75.0, dy=75.0, depth=3.0, dt=dt, velocity=0.3,
                                  breathing=0.1, diffusion_coefficient=0.1, period_steps=24, n_dry=nx * ny // 200)


def oracle_states(args):
    """(steps, n) oracle concentrations of constituent 0 after every step."""
    nx, ny, dt = args
    import clearwater_riverine_amd as cw
    import cwr_oracle as oracle
    from util import oracle_run
    mesh = band(nx, ny, dt)
    inputs1 = cw.synthetic.boundary_input_array(mesh, 1, inlet_period_s=86400.0)
    oracle.derive_coefficients(mesh)
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ref = oracle_run(mesh, inputs1, steps)
    n = mesh['nreal'] + 1
    return (nx, ny, dt), np.array(ref.constituent_dict['c0'].state[1:steps + 1, :n]), time.time() - t0


def main():
    want = {}
    jobs = [(nx, ny, dt) for (nx, ny) in SIZES for dt in DTS if nx * ny <= ORACLE_MAX + 1000]
    if jobs:
        import multiprocessing as mp
        t0 = time.time()
        with mp.get_context('spawn').Pool(min(len(jobs), int(os.environ.get('PROBE_WORKERS', '12')))) as pool:   # (children never touch the GPU)
            for key, states, sec in pool.imap_unordered(oracle_states, jobs):
                want[key] = states
        print(f'# oracle: {len(jobs)} runs of {steps} steps in {time.time() - t0:.0f} s (pool)', flush=True)
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.distributed import PartitionedTransport
    for (nx, ny) in SIZES:
        for dt in DTS:
            mesh = band(nx, ny, dt)
            n = mesh['nreal'] + 1
            for K in KS:
                inputs3 = cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=86400.0)
                scale = np.arange(K) + 1.0
                pt = PartitionedTransport(mesh, inputs3, 0, 1)
                ms, sw, its, flags, kern = [], [], 0, 0, set()
                ref = want.get((nx, ny, dt))
                worst_max, worst_ew = 0.0, 0.0
                try:
                    F = pt.engine.error_factors()[:steps]
                    with warnings.catch_warnings():
                        warnings.simplefilter('ignore')
                        for t in range(steps):
                            t0 = time.perf_counter()
                            r = pt.engine.step(t, tol=1e-12)
                            pt.engine.synchronize()
                            ms.append((time.perf_counter() - t0) * 1e3)
                            sw.append(r.sweeps); its = max(its, r.iterations); flags |= r.flags; kern.add(r.sweep_kernel)
                            if ref is not None:
                                got = pt.gather_state()
                                b = ref[t][:, None] * scale[None, :]
                                peak = np.max(np.abs(b))
                                err = np.abs(got - b)
                                worst_max = max(worst_max, float(err.max() / peak))
                                worst_ew = max(worst_ew, float(np.max(err / (1e-6 * np.abs(b) + 1e-12 * peak))))
                    note = ''
                    if max(sw[2:]) > 2.0 * np.median(sw[2:]) or its > 0 or flags or worst_ew > 1.0 or worst_max > 1e-9:
                        note = '   <<< LOOK'
                    orc = f', vs oracle: max-norm {worst_max:.1e} element-wise {worst_ew:.2g} x bar' if ref is not None else ''
                    print(f'{nx}x{ny} n={n} dt={dt:g} K={K}: median {np.median(ms[2:]):.3f} max {max(ms[2:]):.3f} ms/step, sweeps {min(sw[2:])}-{max(sw[2:])}, bicgstab {its}, '
                          f'flags {flags}, kernel {sorted(kern)}, F {F.min():.3g}-{F.max():.3g}{orc}{note}', flush=True)
                except Exception as exc:
                    print(f'{nx}x{ny} dt={dt:g} K={K}: {type(exc).__name__}: {str(exc)[:160]}   <<< LOOK', flush=True)
                pt.engine.close()


if __name__ == '__main__':
    main()
