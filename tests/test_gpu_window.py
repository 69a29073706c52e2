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
