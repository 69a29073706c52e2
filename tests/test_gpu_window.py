"""Windowed flow-field residency (SURVEY 8 f-1 "time-series streaming"; VERDICT r04 task 4): cwr_flow_window_open / _load.

The reference derives its coefficients per level (utilities.py:513-541), its reader windows a file by datetime_range
(io/hdf.py:149-191) and its own fixture has 10 801 stamps; the all-resident engine holds every level in HBM.  Here a ring of W
levels is refilled one level per step on the engine's flow stream, beside the steps, and the results must be those of the
all-resident engine BIT FOR BIT (deterministic passes): the same coefficients, the same ||J||_inf, the same error factors, the same
zero-coefficient flags, hence the same stop decisions.
"""
import numpy as np
import pytest

import cwr_oracle as oracle
from util import load_plan, multi_inputs, oracle_run, rel_err
from test_gpu_parity import make_engine

pytestmark = pytest.mark.gpu


def windowed_engine(mesh, inputs3, W, first=None):
    import clearwater_riverine_amd as cw
    n = mesh['nreal'] + 1
    eng = cw.TransportEngine(mesh['edges_face1'], mesh['edges_face2'], len(mesh['face_x']), inputs3.shape[2])
    T = len(mesh['dt'])
    eng.flow_window_open(T, W, mesh['dt'], mesh['face_to_face_dist'], mesh['diffusion_coefficient'])
    eng.load_boundary(inputs3[:, n:, :])
    hi = min(T, W if first is None else first)
    eng.flow_window_load(0, mesh['face_flow'][:hi], mesh['edge_velocity'][:hi], mesh['volume'][:hi])
    return eng, hi


def run_pair(mesh, inputs3, W, steps, det=True, check_every=1, chunk=1):
    """Resident and windowed engine side by side; the window is refilled `chunk` levels at a time as soon as their slots are free."""
    n = mesh['nreal'] + 1
    T = len(mesh['dt'])
    res = make_engine(mesh, inputs3)
    win, hi = windowed_engine(mesh, inputs3, W)
    res.set_state(inputs3[0, :n, :]); win.set_state(inputs3[0, :n, :])
    for t in range(steps):
        # levels < t are no longer read by any step: their slots take levels up to t + W - 1
        target = min(T, t + W)
        while hi < target and (target - hi >= chunk or hi < t + 2 or target == T):
            m = min(chunk, target - hi)
            win.flow_window_load(hi, mesh['face_flow'][hi:hi + m], mesh['edge_velocity'][hi:hi + m], mesh['volume'][hi:hi + m])
            hi += m
        ra = res.step(t, tol=1e-12, deterministic=det)
        rb = win.step(t, tol=1e-12, deterministic=det)
        assert (ra.sweeps, ra.iterations, ra.flags, ra.sweep_kernel) == (rb.sweeps, rb.iterations, rb.flags, rb.sweep_kernel), (t, ra, rb)
        if t % check_every == 0 or t == steps - 1:
            a, b = res.get_state(), win.get_state()
            assert np.array_equal(a, b, equal_nan=True), f'step {t}: windowed state differs from the resident one'
            fa, fb = res.get_mass_flux(), win.get_mass_flux()
            assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(fa, fb))
    jn_r, jn_w = res.jacobi_norms(), win.jacobi_norms()
    assert np.array_equal(jn_r[:steps], jn_w[:steps])
    assert np.array_equal(res.error_factors()[:steps], win.error_factors()[:steps])
    out = res.get_state()
    res.close(); win.close()
    return out


def test_plan01_fixture_through_a_16_level_window_equals_the_resident_run_bit_for_bit(gpu_lib):
    """The reference's own 64-level HDF fixture (tests/data/simple_test_cases/plan01_10x5 cut to tests/golden), K = 3, W = 16: every
    step's state and fluxes bitwise, and the oracle's answer at the end."""
    mesh, inp, _ = load_plan('plan01', 0.01)
    K = 3
    inputs3 = multi_inputs(inp, K, seed=3)
    T = inputs3.shape[0]
    assert T == 64
    out = run_pair(mesh, inputs3, 16, T - 1)
    n = mesh['nreal'] + 1
    ref = oracle_run(mesh, inputs3, T - 1)
    for k in range(K):
        assert rel_err(out[:, k], ref.constituent_dict[f'c{k}'].state[T - 1]) <= 1e-9


@pytest.mark.parametrize('K,chunk', [(16, 1), (4, 5)])
def test_200k_cells_60_levels_through_a_16_level_window_equal_the_resident_run_bit_for_bit(gpu_lib, K, chunk):
    """A mesh large enough for the tiled, chained passes (deterministic: walked between two vectors) and for the internal face order
    to matter: 200 k cells x 60 levels through W = 16, one level per step (K = 16) or five at a time (K = 4)."""
    import clearwater_riverine_amd as cw
    steps = 59
    mesh = cw.synthetic.make_mesh(500, 400, steps, seed=8, n_merge=10000, dt=40.0, diffusion_coefficient=0.5)
    oracle.derive_coefficients(mesh)
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=8)
    assert len(mesh['dt']) == 60 and mesh['nreal'] + 1 == 190000
    run_pair(mesh, inputs3, 16, steps, check_every=10, chunk=chunk)


def test_a_windowed_field_with_dry_cells_takes_the_row_wise_factor_at_the_step(gpu_lib, monkeypatch):
    """Where ||J||_inf admits no bound (30 % dry cells) the windowed engine runs the Neumann sweeps when the step comes: the same
    factors as the resident engine, no flag, the same bits."""
    import clearwater_riverine_amd as cw
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    K, steps = 3, 6
    nx, ny = 90, 40
    mesh = cw.synthetic.make_mesh(nx=nx, ny=ny, n_steps=steps, seed=12, n_merge=nx * ny // 25, n_dry=int(0.3 * nx * ny), dt=30.0,
                                  diffusion_coefficient=0.5)
    oracle.derive_coefficients(mesh)
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=12)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        run_pair(mesh, inputs3, 3, steps)


def test_window_errors_and_out_of_order_loads(gpu_lib):
    import clearwater_riverine_amd as cw
    mesh, inp, _ = load_plan('plan01', 0.01)
    inputs3 = multi_inputs(inp, 1)
    n = mesh['nreal'] + 1
    eng, hi = windowed_engine(mesh, inputs3, 4, first=2)          # levels 0, 1 only
    eng.set_state(inputs3[0, :n, :])
    eng.step(0)
    with pytest.raises(IndexError, match='window'):
        eng.step(1)                                              # level 2 has not been loaded
    with pytest.raises(ValueError):
        eng.flow_window_load(2, mesh['face_flow'][2:8], mesh['edge_velocity'][2:8], mesh['volume'][2:8])   # more levels than the ring holds
    # out of order: level 3 first, then 2 -- step 1 and step 2 become possible with the second load
    eng.flow_window_load(3, mesh['face_flow'][3:4], mesh['edge_velocity'][3:4], mesh['volume'][3:4])
    with pytest.raises(IndexError, match='window'):
        eng.step(1)
    eng.flow_window_load(2, mesh['face_flow'][2:3], mesh['edge_velocity'][2:3], mesh['volume'][2:3])
    eng.step(1); eng.step(2)
    # level 4 replaces level 0: step 0 is no longer possible, coefficients of a resident level can be read back
    eng.flow_window_load(4, mesh['face_flow'][4:5], mesh['edge_velocity'][4:5], mesh['volume'][4:5])
    with pytest.raises(IndexError, match='window'):
        eng.get_coefficients(0)
    adv, dif = eng.get_coefficients(4)
    assert np.array_equal(adv, mesh['advection_coeff'][4]) and np.array_equal(dif, mesh['coeff_to_diffusion'][4])
    res = make_engine(mesh, inputs3)
    res.set_state(inputs3[0, :n, :])
    for t in range(3):
        res.step(t)
    assert np.array_equal(res.get_state(), eng.get_state(), equal_nan=True)
    res.close(); eng.close()


def test_facade_flow_window_keyword_and_automatic_choice(gpu_lib, monkeypatch):
    """ClearwaterRiverine(flow_window=W) refills the ring in update(); with CWR_FLOW_RESIDENT_LIMIT_MB too small for the field the
    facade windows by itself.  Same histories as the resident facade, bit for bit (deterministic=True)."""
    import clearwater_riverine_amd as cw
    K, steps = 2, 20
    mesh = cw.synthetic.make_mesh(120, 60, steps, seed=5, n_merge=200, dt=40.0, diffusion_coefficient=0.5)
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=5)
    arrays = {f'c{k}': inputs3[:, :, k].copy() for k in range(K)}
    models = {}
    for label, kw in (('resident', {}), ('window', {'flow_window': 5}), ('auto', {})):
        if label == 'auto':
            monkeypatch.setenv('CWR_FLOW_RESIDENT_LIMIT_MB', '1')
        mdl = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={k: v.copy() for k, v in arrays.items()}, deterministic=True, **kw)
        for _ in range(steps):
            mdl.update()
        models[label] = mdl
    assert models['resident']._flow_window is None and models['window']._flow_window == 5 and models['auto']._flow_window is not None
    assert models['auto']._flow_window < steps + 1
    for label in ('window', 'auto'):
        for nm in arrays:
            assert np.array_equal(models['resident'].mesh[nm], models[label].mesh[nm], equal_nan=True), (label, nm)
            assert np.array_equal(models['resident'].constituent_dict[nm].total_mass_flux, models[label].constituent_dict[nm].total_mass_flux, equal_nan=True)
    for mdl in models.values():
        mdl.close_output(); mdl.engine.close()


def test_window_refills_beside_a_busy_chip_equal_the_resident_run_bit_for_bit(gpu_lib):
    """The ring's refills run on a stream of their own, ordered with the steps by events only -- the kind of code that holds on an
    idle chip and breaks under load.  The 190 k-cell comparison again (one level per step through W = 8) while a second engine in
    another thread keeps the CUs and the copy engines busy (steps + state uploads)."""
    import threading
    import clearwater_riverine_amd as cw
    steps = 39
    mesh = cw.synthetic.make_mesh(500, 400, steps, seed=8, n_merge=10000, dt=40.0, diffusion_coefficient=0.5)
    oracle.derive_coefficients(mesh)
    inputs3 = cw.synthetic.distinct_input_array(mesh, 4, seed=8)
    other = cw.synthetic.make_mesh(300, 300, 4, seed=9, n_merge=2000, dt=40.0, diffusion_coefficient=0.5)
    oracle.derive_coefficients(other)
    other_in = cw.synthetic.distinct_input_array(other, 8, seed=9)
    stop = threading.Event()
    errors = []

    def load():
        try:
            n = other['nreal'] + 1
            eng = make_engine(other, other_in)
            while not stop.is_set():
                eng.set_state(other_in[0, :n, :])
                for t in range(3):
                    eng.step(t, tol=1e-12)
            eng.close()
        except Exception as exc:                      # pragma: no cover
            errors.append(exc)

    th = threading.Thread(target=load)
    th.start()
    try:
        run_pair(mesh, inputs3, 8, steps, check_every=6)
    finally:
        stop.set(); th.join()
    assert not errors, errors


@pytest.mark.parametrize('K,nx,ny,W', [(3, 120, 60, 6), (2, 40, 30, 16)])
def test_facade_streams_a_lazy_level_source_through_the_ring_bit_for_bit(gpu_lib, tmp_path, K, nx, ny, W):
    """file -> staging -> ring.  The three (T, .) arrays live in an .npz-backed store on disk (one .npy per array, memory-
    mapped) behind a LEVEL SOURCE -- a callable (t0, t1) -> the levels -- and the facade (flow_window=W) pulls W / 2 levels at a time into two
    page-locked staging blocks (levels.FlowWindowFeeder), the boundary values of the same levels with them (cwr_boundary_window_load; the input
    arrays are SparseInputArrays: no (T, ncell) array anywhere).  Histories equal to the resident facade's BIT FOR BIT; every level read once,
    never more than W / 2 at a time.  K = 3 is carried as 4 (the boundary rows are padded on the flow stream); 7 200 cells take the tiled
    passes with a renumbered engine (volumes permuted in the staging block), 1 200 the one-launch solver."""
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.model import SparseInputArray
    steps = 30
    mesh = cw.synthetic.make_mesh(nx, ny, steps, seed=5, n_merge=nx * ny // 36, dt=40.0, diffusion_coefficient=0.5)
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=5)
    T, ncell = inputs3.shape[0], inputs3.shape[1]
    n = mesh['nreal'] + 1
    assert not inputs3[1:, :n].any()                        # (this input family has no real-cell entries behind the initial row)
    names = [f'c{k}' for k in range(K)]
    # (two boundary-condition lines over the ghost faces: _mass_bal_global of the STREAMED run must equal the resident one -- the feeder keeps
    # the lines' face flows as the chunks pass by)
    gf = np.nonzero(np.asarray(mesh['edges_face2']) > mesh['nreal'])[0]
    bfaces = {'US_Flow': gf[0::2], 'DS_Stage': gf[1::2]}
    res_mesh = cw.Mesh(dict(mesh)); res_mesh.attrs['boundary_faces'] = bfaces
    res = cw.ClearwaterRiverine(mesh=res_mesh, input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)}, deterministic=True)
    for _ in range(steps):
        res.update()
    # the store: one .npy per array, opened memory-mapped (np.load(mmap_mode='r')): reading a slice touches only its pages
    for key in ('face_flow', 'edge_velocity', 'volume'):
        np.save(tmp_path / f'{key}.npy', np.ascontiguousarray(mesh[key], dtype=np.float32))
    maps = [np.load(tmp_path / f'{key}.npy', mmap_mode='r') for key in ('face_flow', 'edge_velocity', 'volume')]
    calls = []

    def source(t0, t1):
        calls.append((t0, t1))
        return tuple(np.asarray(m[t0:t1]) for m in maps)

    lazy_mesh = {k: v for k, v in mesh.items() if k not in ('face_flow', 'edge_velocity', 'volume', 'advection_coeff', 'coeff_to_diffusion')}
    lazy_mesh = cw.Mesh(lazy_mesh); lazy_mesh.attrs['boundary_faces'] = bfaces
    lazy_mesh['level_source'] = source
    ghosts = np.arange(n, ncell)
    sparse = {nm: SparseInputArray(T, ncell, np.where(np.arange(ncell) < n, inputs3[0, :, k], 0.0), ghosts, inputs3[:, n:, k]) for k, nm in enumerate(names)}
    win = cw.ClearwaterRiverine(mesh=lazy_mesh, input_arrays=sparse, deterministic=True, flow_window=W)
    assert win._feeder is not None and win._flow_window == W
    for _ in range(steps):
        win.update()
        assert (win.last_step.sweeps, win.last_step.flags) == (res.last_step.sweeps, res.last_step.flags) or win.time_step < steps
    assert sum(b - a for a, b in calls) == T and max(b - a for a, b in calls) == W // 2
    for nm in names:
        assert np.array_equal(res.mesh[nm], win.mesh[nm], equal_nan=True), nm
        assert np.array_equal(res.constituent_dict[nm].total_mass_flux, win.constituent_dict[nm].total_mass_flux, equal_nan=True)
    assert np.array_equal(res.engine.jacobi_norms()[:steps], win.engine.jacobi_norms()[:steps])
    a, b = res.mass_bal_global(names[0]), win.mass_bal_global(names[0])
    assert list(a) == list(b)
    for key in a:
        assert (np.isnan(a[key]) and np.isnan(b[key])) or a[key] == b[key] or np.isclose(a[key], b[key], rtol=1e-12, atol=0.0), (key, a[key], b[key])
    for mdl in (res, win):
        mdl.close_output(); mdl.engine.close()


def test_boundary_levels_noted_without_flow_levels_and_on_resident_engines(gpu_lib):
    """cwr_boundary_window_load on its own: (i) a windowed engine whose flow levels are all in the ring already and whose boundary values arrive
    level by level just in time (the step flushes and waits for them), (ii) an engine with a resident flow field (a blocking upload).  Same bits
    as the engine that was given all levels at once."""
    mesh, inp, _ = load_plan('plan01', 0.01)
    K = 2
    inputs3 = multi_inputs(inp, K, seed=4)
    n = mesh['nreal'] + 1
    T, steps = inputs3.shape[0], 12
    ref = make_engine(mesh, inputs3)
    ref.set_state(inputs3[0, :n, :])
    win, hi = windowed_engine(mesh, inputs3, T)                  # every flow level resident in a "ring" of T
    win.alloc_boundary(T)                                        # (replaces the values windowed_engine loaded: zeros)
    res = make_engine(mesh, inputs3)
    res.alloc_boundary(T)
    win.set_state(inputs3[0, :n, :]); res.set_state(inputs3[0, :n, :])
    win.boundary_window_load(0, inputs3[0:1, n:, :]); res.boundary_window_load(0, inputs3[0:1, n:, :])
    for t in range(steps):
        g = np.ascontiguousarray(inputs3[t + 1:t + 2, n:, :])   # the level step t reads, handed over just before it
        win.boundary_window_load(t + 1, g); res.boundary_window_load(t + 1, g)
        a, b, c = ref.step(t, deterministic=True), win.step(t, deterministic=True), res.step(t, deterministic=True)
        assert a.sweeps == b.sweeps == c.sweeps
    sa, sb, sc = ref.get_state(), win.get_state(), res.get_state()
    assert np.array_equal(sa, sb, equal_nan=True) and np.array_equal(sa, sc, equal_nan=True)
    with pytest.raises((IndexError, ValueError)):
        win.boundary_window_load(T - 1, inputs3[0:2, n:, :])     # beyond the allocated levels
    for e in (ref, win, res):
        e.close()
