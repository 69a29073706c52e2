"""Round 4: a seeded soak of the whole step against the oracle (scipy spsolve per constituent) over random small meshes -- sizes, cell
shapes (6- and 8-sided cells, dry cells), time steps from CFL 0.6 to CFL 200, constituent counts, grid caps that make small meshes chain,
numberings, deterministic / in-place passes, bent channels.  Every case: 2 steps, max-norm <= 1e-9 AND the element-wise bar of tests/util.py.
usage: python tests/soak_step.py [n_cases] [seed]      (tests/test_gpu_soak.py runs 40 cases; 250 cases of seed 777 take 45 s on one MI355X)"""
import os, sys, time, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'oracle')); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import warnings
import numpy as np
import clearwater_riverine_amd as cw
import cwr_oracle as oracle
from util import oracle_run, rel_err
from clearwater_riverine_amd.distributed import PartitionedTransport

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
# (CWR_SMALL_MAX_CELLS: the test runner's conftest sets it to 0 for tests written on the tiled passes; the soak takes what the engine takes by
# default -- meshes of 4 097 ... 24 576 cells through the several-workgroups solver unless CWR_NO_SMALL is drawn: VERDICT r05 weak 8)
KNOBS = ['CWR_TCL_GRID', 'CWR_TILE_ORDER', 'CWR_LANE_KIND', 'CWR_NO_SMALL', 'CWR_DET_DEFAULT_K', 'CWR_CHAIN_MIN_TILES', 'CWR_SMALL_MAX_CELLS']
bad = 0
t_all = time.time()
for case in range(n_cases):
    for k in KNOBS: os.environ.pop(k, None)
    nx = int(rng.integers(24, 200)); ny = int(rng.integers(16, 120))
    K = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16]))
    dt = float(rng.choice([10.0, 40.0, 40.0, 400.0, 3600.0]))
    nb = nx * ny
    n_merge = int(rng.choice([0, nb // 40, nb // 20])); n_merge4 = int(rng.choice([0, 0, nb // 200]))
    n_dry = int(rng.choice([0, 0, 2, nb // 50]))
    bend = float(rng.choice([0.0, 0.0, 0.6])) if nx >= 3 * ny else 0.0
    env = {}
    if rng.random() < 0.6: env['CWR_TCL_GRID'] = str(int(rng.choice([8, 16, 32, 64])))
    if rng.random() < 0.5: env['CWR_TILE_ORDER'] = str(rng.choice(['lanes', 'hilbert']))
    if rng.random() < 0.3: env['CWR_LANE_KIND'] = str(rng.choice(['straight', 'channel']))
    if rng.random() < 0.5: env['CWR_NO_SMALL'] = '1'
    if rng.random() < 0.3: env['CWR_DET_DEFAULT_K'] = str(rng.choice(['0', '99']))
    if rng.random() < 0.3: env['CWR_CHAIN_MIN_TILES'] = '1'
    det = bool(rng.random() < 0.3)
    steps = 2
    desc = f'case {case}: {nx}x{ny} K={K} dt={dt:g} merge={n_merge}/{n_merge4} dry={n_dry} bend={bend} det={det} {env}'
    try:
        mesh = cw.synthetic.make_mesh(nx, ny, steps, seed=int(rng.integers(1, 10**6)), n_merge=n_merge, n_merge4=n_merge4, n_dry=n_dry, dt=dt,
                                      diffusion_coefficient=float(rng.choice([0.1, 0.5, 2.0])), breathing=0.0 if dt > 400 else 0.02)
        if bend > 0:
            mesh = cw.synthetic.bend_channel(mesh, bend)
        inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=int(rng.integers(1, 1000)))
        oracle.derive_coefficients(mesh)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            ref = oracle_run(mesh, inputs3, steps)
        n = mesh['nreal'] + 1
        want = np.stack([ref.constituent_dict[f'c{k}'].state[steps, :n] for k in range(K)], axis=1)
        os.environ.update(env)
        pt = PartitionedTransport(mesh, inputs3, 0, 1)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            rs = [pt.step(t, tol=1e-12, mass_flux=True, deterministic=det) for t in range(steps)]
        err = rel_err(pt.gather_state(), want)
        ok = err <= 1e-9
        print(f'{"ok " if ok else "BAD"} {desc}: n={n} kernel {rs[-1].sweep_kernel} chained {rs[-1].chained} sweeps {rs[-1].sweeps} bicg {rs[-1].iterations} flags {rs[-1].flags} err {err:.1e}', flush=True)
        bad += 0 if ok else 1
        pt.engine.close()
    except Exception as ex:                                                  # (a soak reports and goes on)
        bad += 1
        print(f'EXC {desc}: {type(ex).__name__}: {str(ex)[:300]}', flush=True)
        traceback.print_exc(limit=3)
print(f'{n_cases - bad} of {n_cases} cases ok in {time.time() - t_all:.0f} s')
sys.exit(1 if bad else 0)
