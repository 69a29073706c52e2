"""GPU tests of the chained in-place passes (round 3): tile chains along the flow, relaxed in place by the persistent grid of
the tiled J^2 pass -- block Gauss-Seidel along the flow in which no block waits for another.  What replaces spsolve
(transport.py:249) must reach spsolve's answer whatever the schedule; the schedule only decides how many passes that takes.

The engine builds schedules from four tiles per block up (1 M cells x 16 constituents: 15 per block); CWR_TCL_GRID caps the
grid of the tiled pass so that a 40 000-cell test mesh gets lists of that length."""
import numpy as np
import pytest

import cwr_oracle as oracle
from util import oracle_run, rel_err

pytestmark = pytest.mark.gpu

K = 16
GRID = 64


def case(steps=3, dt=40.0, nx=200, ny=200):
    import clearwater_riverine_amd as cw
    mesh = cw.synthetic.make_mesh(nx, ny, steps, seed=11, n_merge=nx * ny // 20, dt=dt, diffusion_coefficient=0.5)
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=2)
    return mesh, inputs3


def transport(mesh, inputs3, monkeypatch, chains=True, **env):
    from clearwater_riverine_amd.distributed import PartitionedTransport
    monkeypatch.setenv('CWR_TCL_GRID', str(GRID))
    if chains:
        monkeypatch.delenv('CWR_NO_CHAINS', raising=False)
    else:
        monkeypatch.setenv('CWR_NO_CHAINS', '1')
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    return PartitionedTransport(mesh, inputs3, 0, 1)


def test_engine_schedule_equals_the_numpy_specification(gpu_lib, monkeypatch):
    """build_chain_schedule (csrc/cwr_engine.hip: link fluxes on the device, chains and lists on the host) against
    schedule.py, the numpy statement of the same construction, on the engine's own tiling."""
    from clearwater_riverine_amd import schedule as sch
    mesh, inputs3 = case()
    pt = transport(mesh, inputs3, monkeypatch)
    eng, lm = pt.engine, pt.local
    ready, ntiles, grid, TR = eng.tiling_info()
    assert ready and grid == GRID and ntiles >= 4 * grid and TR == 64
    assert eng.get_tile_schedule()[0] is None                  # nothing built before the first step
    r = pt.step(0, tol=1e-12)
    got, level, built = eng.get_tile_schedule()
    assert level == 0 and built == 1 and r.sweep_kernel == 6
    adv = eng.get_coefficients(0)[0]                           # advection_coeff[0] in the faces this engine was created with
    # (one stream per block: with the column reuse of round 3 a tile's chain successor simply comes next in the block's list)
    want = sch.chain_schedule(lm.face1, lm.face2, adv, lm.n_rows, TR, ntiles, grid, streams_per_block=1)
    assert got.shape == want.shape and np.array_equal(got, want)
    tiles = got[got >= 0]
    assert len(tiles) == ntiles and np.array_equal(np.sort(tiles), np.arange(ntiles))     # every tile exactly once
    assert (np.diff((got >= 0).astype(int), axis=0) <= 0).all()                            # lists are dense prefixes
    eng.close()


def test_chained_passes_reach_the_direct_solve_in_fewer_sweeps(gpu_lib, monkeypatch):
    """Three steps against the oracle (spsolve), element-wise, with and without chains; the chained run must not need more
    sweeps, and the two HIP runs agree far inside the tolerance."""
    steps = 3
    mesh, inputs3 = case(steps)
    oracle.derive_coefficients(mesh)
    ref = oracle_run(mesh, inputs3, steps)
    n = mesh['nreal'] + 1
    want = np.stack([ref.constituent_dict[f'c{k}'].state[steps, :n] for k in range(K)], axis=1)
    out, sweeps = {}, {}
    for chains in (False, True):
        pt = transport(mesh, inputs3, monkeypatch, chains)
        rs = [pt.step(t, tol=1e-12) for t in range(steps)]
        assert all(r.sweep_kernel == 6 and r.flags == 0 and r.max_rel_residual <= 1e-12 for r in rs)
        assert (pt.engine.get_tile_schedule()[0] is not None) == chains
        out[chains], sweeps[chains] = pt.gather_state(), [r.sweeps for r in rs]
        pt.engine.close()
        assert rel_err(out[chains], want) <= 1e-9
    assert rel_err(out[True], out[False]) <= 1e-10
    assert all(c <= p for c, p in zip(sweeps[True], sweeps[False])), sweeps
    assert sum(sweeps[True]) < 0.95 * sum(sweeps[False]), sweeps


def test_a_schedule_against_the_flow_costs_sweeps_not_correctness(gpu_lib, monkeypatch):
    """No block waits for another, so ANY complete schedule is safe: the lists of the reversed flow field (every chain walked
    upstream: the worst order) and a random permutation of the tiles still reach the oracle's answer."""
    from clearwater_riverine_amd import schedule as sch
    mesh, inputs3 = case(2)
    oracle.derive_coefficients(mesh)
    ref = oracle_run(mesh, inputs3, 2)
    n = mesh['nreal'] + 1
    want = np.stack([ref.constituent_dict[f'c{k}'].state[2, :n] for k in range(K)], axis=1)
    rng = np.random.default_rng(5)
    for kind in ('reversed', 'random'):
        pt = transport(mesh, inputs3, monkeypatch)
        eng, lm = pt.engine, pt.local
        _, ntiles, grid, TR = eng.tiling_info()
        if kind == 'reversed':
            adv = -np.asarray(mesh['face_flow'][0])[lm.edge_global]
            sc = sch.chain_schedule(lm.face1, lm.face2, adv, lm.n_rows, TR, ntiles, grid)
        else:
            perm = rng.permutation(ntiles)
            depth = -(-ntiles // grid)
            sc = np.full((depth, grid), -1, np.int32)
            for b in range(grid):
                lst = perm[b::grid]
                sc[:len(lst), b] = lst
        eng.set_tile_schedule(sc)
        for t in range(2):
            assert pt.step(t, tol=1e-12).flags == 0
        got, level, built = eng.get_tile_schedule()
        assert level == -1 and built == 0 and np.array_equal(got, sc)        # the caller's schedule is never replaced
        assert rel_err(pt.gather_state(), want) <= 1e-9
        eng.close()


def test_schedule_is_rebuilt_as_the_levels_advance_and_bad_schedules_are_refused(gpu_lib, monkeypatch):
    mesh, inputs3 = case(5)
    pt = transport(mesh, inputs3, monkeypatch, CWR_CHAIN_REFRESH='2')
    eng = pt.engine
    for t in range(5):
        pt.step(t, tol=1e-10)
    _, level, built = eng.get_tile_schedule()
    assert built == 3 and level == 4                                         # levels 0, 2, 4
    _, ntiles, grid, _ = eng.tiling_info()
    depth = -(-ntiles // grid)
    sc = np.full((depth, grid), -1, np.int32)
    for b in range(grid):
        lst = np.arange(ntiles)[b::grid]
        sc[:len(lst), b] = lst
    bad = sc.copy(); bad[0, 0] = bad[0, 1]                                   # a tile twice, another one missing
    with pytest.raises(ValueError, match='twice|exactly once'):
        eng.set_tile_schedule(bad)
    with pytest.raises(ValueError, match='n_lists'):
        eng.set_tile_schedule(sc[:, :grid - 8])
    hole = sc.copy(); hole[0, 3], hole[depth - 1, 3] = -1, sc[0, 3]          # a list that goes on behind its end
    with pytest.raises(ValueError):
        eng.set_tile_schedule(hole)
    eng.set_tile_schedule(sc)                                                # a complete one is taken
    eng.close()


def test_without_chains_runs_are_bitwise_reproducible_and_with_them_within_the_tolerance(gpu_lib, monkeypatch):
    """The ping-pong passes read only what the previous launch wrote: bitwise reproducible run to run (CWR_NO_CHAINS=1).
    In a chained pass a tile may or may not see a neighbouring chain's update of the same launch (the blocks do not wait
    for each other), so two runs agree to the solver tolerance, not bit for bit."""
    mesh, inputs3 = case(2)
    runs = {}
    for chains in (False, True):
        outs = []
        for _ in range(2):
            pt = transport(mesh, inputs3, monkeypatch, chains)
            for t in range(2):
                pt.step(t, tol=1e-12)
            outs.append(pt.gather_state())
            pt.engine.close()
        runs[chains] = outs
    assert np.array_equal(runs[False][0], runs[False][1])
    assert rel_err(runs[True][0], runs[True][1]) <= 1e-10


def test_the_deterministic_step_flag_selects_the_deterministic_passes_per_step(gpu_lib, monkeypatch):
    """regression-stable output without an environment variable.  On ONE engine that chains by default, steps taken with
    deterministic=True (CWR_STEP_DETERMINISTIC) run the passes between two vectors (the tile chains still walked: chained == 2) -- cwr_step_info.chained says which ran -- and two engines
    driven that way agree bit for bit, while the default steps of the same engines agree to the solver tolerance only.  The facade
    forwards its constructor keyword."""
    import clearwater_riverine_amd as cw
    mesh, inputs3 = case(3)
    outs = {True: [], False: []}
    for det in (True, False):
        for _ in range(2):
            pt = transport(mesh, inputs3, monkeypatch, CWR_TILE_ORDER='lanes')
            rs = [pt.step(t, tol=1e-12, deterministic=det) for t in range(3)]
            assert all(r.sweep_kernel == 6 and r.chained == (2 if det else 1) for r in rs), [(r.sweep_kernel, r.chained) for r in rs]
            outs[det].append(pt.gather_state())
            # the flag is per step: the other kind of pass on the same engine, same answer to the tolerance
            pt.engine.set_state(inputs3[0, :mesh['nreal'] + 1, :])
            r = pt.step(0, tol=1e-12, deterministic=not det)
            assert r.chained == (1 if det else 2)
            pt.engine.close()
    assert np.array_equal(outs[True][0], outs[True][1])
    assert rel_err(outs[False][0], outs[False][1]) <= 1e-10 and rel_err(outs[False][0], outs[True][0]) <= 1e-10
    # round 4: the deterministic passes of a single engine walk the same lists along the flow (a tile takes its predecessor's rows
    # fresh from LDS, everything else from the pass's read-only input): fewer sweeps than in tile order (CWR_DET_WALK=0), same answer
    sweeps = {}
    for walk in ('0', '1'):                                      # (ends with the default for the facade below)
        pt = transport(mesh, inputs3, monkeypatch, CWR_TILE_ORDER='lanes', CWR_DET_WALK=walk)
        rs = [pt.step(t, tol=1e-12, deterministic=True) for t in range(3)]
        assert all(r.chained == (2 if walk == '1' else 0) for r in rs)
        sweeps[walk] = sum(r.sweeps for r in rs)
        assert rel_err(pt.gather_state(), outs[True][0]) <= 1e-10
        if walk == '1':
            assert np.array_equal(pt.gather_state(), outs[True][0])
        pt.engine.close()
    assert sweeps['1'] < sweeps['0'], sweeps
    names = [f'c{k}' for k in range(K)]
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)},
                                  deterministic=True)
    model.update()
    assert model.last_step.chained == 2 and model.last_step.sweep_kernel == 6
    model.engine.close()


@pytest.mark.parametrize('Kc,grid', [(1, 32), (3, 32), (12, 32)])
def test_chained_passes_at_other_constituent_counts_match_the_oracle(gpu_lib, monkeypatch, Kc, grid):
    """The lane mappings of the tiled pass other than K = 16 (one constituent per lane with 256-row tiles, odd K, the four-wide
    mapping at K = 12) through the chained, column-reusing passes: element-wise against spsolve output."""
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.distributed import PartitionedTransport
    steps = 2
    mesh = cw.synthetic.make_mesh(200, 200, steps, seed=12, n_merge=2000, dt=40.0, diffusion_coefficient=0.5)
    inputs3 = cw.synthetic.distinct_input_array(mesh, Kc, seed=3)
    oracle.derive_coefficients(mesh)
    ref = oracle_run(mesh, inputs3, steps)
    n = mesh['nreal'] + 1
    want = np.stack([ref.constituent_dict[f'c{k}'].state[steps, :n] for k in range(Kc)], axis=1)
    monkeypatch.setenv('CWR_TCL_GRID', str(grid))
    monkeypatch.delenv('CWR_NO_CHAINS', raising=False)
    # a single engine with up to 8 constituents takes the deterministic chained passes by default (they cost 1-3.5 % there);
    # CWR_DET_DEFAULT_K=0 gives the in-place passes at every K: both kinds through every lane mapping
    for det_k, kind in (('0', 1), (None, 2 if Kc <= 8 else 1)):
        if det_k is None:
            monkeypatch.delenv('CWR_DET_DEFAULT_K', raising=False)
        else:
            monkeypatch.setenv('CWR_DET_DEFAULT_K', det_k)
        pt = PartitionedTransport(mesh, inputs3, 0, 1)
        rs = [pt.step(t, tol=1e-12) for t in range(steps)]
        sched, _, _ = pt.engine.get_tile_schedule()
        assert sched is not None and all(r.sweep_kernel == 6 and r.flags == 0 and r.chained == kind for r in rs), [(r.sweep_kernel, r.chained) for r in rs]
        assert rel_err(pt.gather_state(), want) <= 1e-9
        pt.engine.close()


def test_flow_that_reverses_mid_run_and_dry_cells(gpu_lib, monkeypatch):
    """A tidal-style field: the through-flow reverses between levels 2 and 3 (the inlet becomes the outlet), two cells are dry.
    The chains are re-derived every level here (CWR_CHAIN_REFRESH=1), turn round with the flow, and every level matches the
    oracle; with the default refresh the stale (now upstream-running) lists still give the right answer, in more sweeps."""
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.distributed import PartitionedTransport
    steps = 5
    mesh = cw.synthetic.make_mesh(200, 200, steps, seed=13, n_merge=2000, n_dry=2, dt=40.0, diffusion_coefficient=0.5, breathing=0.0,
                                  steady=True)
    for key in ('face_flow', 'edge_velocity'):
        mesh[key] = mesh[key].copy()
        mesh[key][3:] *= -1.0                               # (a steady field keeps discrete continuity under a sign flip)
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=6)
    inputs3[:, mesh['outlet_ghost_cells'], :] = 2.5         # every open boundary carries a value whichever way the water runs
    oracle.derive_coefficients(mesh)
    ref = oracle_run(mesh, inputs3, steps)
    n = mesh['nreal'] + 1
    want = np.stack([ref.constituent_dict[f'c{k}'].state[steps, :n] for k in range(K)], axis=1)
    sweeps = {}
    for refresh in ('1', '64'):
        pt = transport(mesh, inputs3, monkeypatch, CWR_CHAIN_REFRESH=refresh)
        # (dry cells break continuity for their neighbours: ||J||_inf > 1 -- which used to clamp the element-wise rule, with a
        # warning; round 4: the rule is scaled by the row-wise bound, finite here, and the steps run clean)
        assert pt.engine.jacobi_norms()[:steps].max() > 1.0 and pt.engine.error_factors()[:steps].max() < 100.0
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('error')
            rs = [pt.step(t, tol=1e-12) for t in range(steps)]
        assert all(r.flags == 0 and r.chained == 1 for r in rs)
        sched0 = pt.engine.get_tile_schedule()
        assert sched0[0] is not None and sched0[2] == (steps if refresh == '1' else 1)
        assert rel_err(pt.gather_state(), want) <= 1e-9
        sweeps[refresh] = [r.sweeps for r in rs]
        pt.engine.close()
    assert sum(sweeps['1'][3:]) <= sum(sweeps['64'][3:]), sweeps      # lists that follow the reversed flow are never worse


@pytest.mark.parametrize('seed', range(8))
def test_random_meshes_flows_and_constituent_counts_through_the_chained_passes(gpu_lib, monkeypatch, seed):
    """Seeded sweep over what the chained, column-reusing passes can meet: mesh aspect (the lane order turns with the flow axis),
    share of 6- and 8-sided cells, dry cells, time step from CFL ~1 to ~60 (2 to 8 tile-local applications), steady / unsteady /
    reversed fields, K in {1, 2, 5, 8, 16, 20}, grid caps that give lists of 3 to 20 tiles.  Every cell against spsolve."""
    import warnings
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.distributed import PartitionedTransport
    rng = np.random.default_rng(1000 + seed)
    Kc = int(rng.choice([1, 2, 5, 8, 16, 20]))
    nx, ny = [(260, 150), (150, 260), (200, 200), (320, 120)][seed % 4]
    dt = float(rng.choice([15.0, 40.0, 200.0, 900.0]))
    steps = 2
    mesh = cw.synthetic.make_mesh(nx, ny, steps, seed=50 + seed, n_merge=int(rng.integers(0, nx * ny // 8)),
                                  n_merge4=int(rng.integers(0, 200)), n_dry=int(rng.integers(0, 3)), dt=dt,
                                  diffusion_coefficient=float(rng.choice([0.05, 0.5, 2.0])), breathing=0.0 if dt > 100 else 0.02,
                                  eddy=float(rng.choice([0.0, 0.3, 0.8])), steady=bool(rng.integers(0, 2)))
    if seed % 3 == 2:                                           # the water runs the other way
        for key in ('face_flow', 'edge_velocity'):
            mesh[key] = -mesh[key]
        if not np.allclose(mesh['volume'][0], mesh['volume'][-1]):
            pytest.skip('a reversed unsteady field breaks continuity (volumes were integrated for the forward field)')
    inputs3 = cw.synthetic.distinct_input_array(mesh, Kc, seed=seed)
    inputs3[:, mesh['outlet_ghost_cells'], :] = 1.5
    oracle.derive_coefficients(mesh)
    ref = oracle_run(mesh, inputs3, steps)
    n = mesh['nreal'] + 1
    want = np.stack([ref.constituent_dict[f'c{k}'].state[steps, :n] for k in range(Kc)], axis=1)
    monkeypatch.setenv('CWR_TCL_GRID', str(int(rng.choice([8, 16, 32]))))
    monkeypatch.delenv('CWR_NO_CHAINS', raising=False)
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)         # (dry cells / CFL 60: the clamp flag, tested elsewhere)
        rs = [pt.step(t, tol=1e-12, max_iter=100000) for t in range(steps)]
    assert all(r.sweep_kernel == 6 for r in rs)
    if pt.engine.get_tile_schedule()[0] is None:
        pytest.skip('lists too short to chain at this K / grid')
    clamped = any(r.flags & cw.engine.INFO_ELEMENTWISE_CLAMPED for r in rs)
    assert rel_err(pt.gather_state(), want, **({'ew_rtol': 1e-4, 'ew_atol': 1e-9} if clamped else {})) <= (1e-6 if clamped else 1e-9)
    pt.engine.close()


def test_chained_passes_on_a_meander_follow_the_channel(gpu_lib, monkeypatch):
    """Round 4: a channel that bends (synthetic.bend_channel: the straight test mesh laid along a sine-generated centre line).  The
    lanes of the internal numbering follow the banks (ordering.channel_coordinates), the engine's chains follow the lanes, and the
    answer is the oracle's whichever numbering ran -- straight lanes, which cut across the bends, need more sweeps for it."""
    import clearwater_riverine_amd as cw
    steps = 3
    mesh = cw.synthetic.make_mesh(400, 96, steps, seed=14, n_merge=1900, dt=40.0, diffusion_coefficient=0.5)
    mesh = cw.synthetic.bend_channel(mesh, 1.0)
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=5)
    oracle.derive_coefficients(mesh)
    ref = oracle_run(mesh, inputs3, steps)
    n = mesh['nreal'] + 1
    want = np.stack([ref.constituent_dict[f'c{k}'].state[steps, :n] for k in range(K)], axis=1)
    sweeps = {}
    for kind in ('straight', 'auto'):
        pt = transport(mesh, inputs3, monkeypatch, CWR_TILE_ORDER='lanes', CWR_LANE_KIND=kind)
        rs = [pt.step(t, tol=1e-12) for t in range(steps)]
        assert all(r.sweep_kernel == 6 and r.chained == 1 and r.flags == 0 for r in rs)
        assert pt.engine.get_tile_schedule()[0] is not None
        assert rel_err(pt.gather_state(), want) <= 1e-9
        sweeps[kind] = sum(r.sweeps for r in rs)
        pt.engine.close()
    assert sweeps['auto'] < sweeps['straight'], sweeps


@pytest.mark.parametrize('K', [4, 16])
def test_deterministic_chained_steps_beside_a_busy_chip_repeat_the_quiet_run_bit_for_bit(gpu_lib, K):
    """A 190 k-cell engine (chained passes walked between two vectors, graph replays, the speculative tail behind the copy-free check)
    stepping with deterministic=True while a second engine in another thread keeps the chip busy: the same sweep counts and bits as
    alone.  (Ordering bugs between streams, events and page-locked notifications hide on an idle chip.)"""
    import threading
    import clearwater_riverine_amd as cw
    from test_gpu_parity import make_engine
    steps = 8
    mesh = cw.synthetic.make_mesh(500, 400, steps, seed=8, n_merge=10000, dt=40.0, diffusion_coefficient=0.5)
    oracle.derive_coefficients(mesh)
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=8)
    other = cw.synthetic.make_mesh(300, 300, 4, seed=9, n_merge=2000, dt=40.0, diffusion_coefficient=0.5)
    oracle.derive_coefficients(other)
    other_in = cw.synthetic.distinct_input_array(other, 8, seed=9)
    n = mesh['nreal'] + 1

    def run():
        eng = make_engine(mesh, inputs3)
        eng.set_state(inputs3[0, :n, :])
        sw = [eng.step(t, tol=1e-12, deterministic=True).sweeps for t in range(steps)]
        out = (sw, eng.get_state(), eng.get_mass_flux()[0])
        eng.close()
        return out

    quiet = run()
    stop = threading.Event()
    errors = []

    def load():
        try:
            m = other['nreal'] + 1
            eng = make_engine(other, other_in)
            while not stop.is_set():
                eng.set_state(other_in[0, :m, :])
                for t in range(3):
                    eng.step(t, tol=1e-12)
                eng.get_state()
            eng.close()
        except Exception as exc:                      # pragma: no cover
            errors.append(exc)

    th = threading.Thread(target=load)
    th.start()
    try:
        busy = [run() for _ in range(2)]
    finally:
        stop.set(); th.join()
    assert not errors, errors
    for b in busy:
        assert b[0] == quiet[0]
        assert np.array_equal(b[1], quiet[1], equal_nan=True) and np.array_equal(b[2], quiet[2], equal_nan=True)


@pytest.mark.parametrize('K', [16, 4, 1])
def test_wave_sliced_entry_layout_gives_the_csr_passes_bit_for_bit(gpu_lib, K, monkeypatch):  # noqa: F811 (K shadows the module's default on purpose)
    """Round 6 A/B knob CWR_TCL_ELL=1: the tiled pass reads its J^2 entries in the wave-sliced (ELLPACK per wave) layout of host::build_ell --
    scalar loop control, constant address increments -- instead of CSR order.  Same sums in the same order (padding adds + 0 x): deterministic
    steps equal the default layout's BIT FOR BIT, sweep counts included; oracle parity on top."""
    import clearwater_riverine_amd as cw
    from test_gpu_parity import make_engine
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    steps = 3
    mesh = cw.synthetic.make_mesh(150, 90, steps, seed=17, n_merge=400, n_merge4=60, n_dry=3, dt=40.0, diffusion_coefficient=0.5)
    oracle.derive_coefficients(mesh)
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=17)
    n = mesh['nreal'] + 1
    out = {}
    for ell in ('0', '1'):
        monkeypatch.setenv('CWR_TCL_ELL', ell)
        eng = make_engine(mesh, inputs3)
        eng.set_state(inputs3[0, :n, :])
        sw = [eng.step(t, tol=1e-12, deterministic=True).sweeps for t in range(steps)]
        out[ell] = (sw, eng.get_state(), eng.get_mass_flux())
        eng.close()
    assert out['0'][0] == out['1'][0]
    assert np.array_equal(out['0'][1], out['1'][1], equal_nan=True)
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out['0'][2], out['1'][2]))
    ref = oracle_run(mesh, inputs3, steps)
    for k in range(K):
        assert rel_err(out['1'][1][:n, k], ref.constituent_dict[f'c{k}'].state[steps][:n]) <= 1e-9
