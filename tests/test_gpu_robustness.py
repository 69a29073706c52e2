"""GPU tests of the solver's acceptance rules and failure behaviour (through the C ABI):
element-wise parity at plume fronts, retry after a failed step, the small-mesh solver's hand-over to BiCGSTAB,
real-cell inputs at levels >= 1 (transport.py:258-264), tolerance flags."""
import numpy as np
import pytest

import cwr_oracle as oracle
from util import flux_err, oracle_run, rel_err
from test_gpu_parity import make_engine

pytestmark = pytest.mark.gpu


def distinct_case(K, **kw):
    import clearwater_riverine_amd as cw
    mesh = cw.synthetic.make_mesh(**kw)
    oracle.derive_coefficients(mesh)
    return mesh, cw.synthetic.distinct_input_array(mesh, K, seed=kw.get('seed', 0))


@pytest.mark.parametrize('path,nx,ny', [('one-launch small-mesh solver', 70, 30), ('tiled block-asynchronous passes', 200, 40)])
@pytest.mark.parametrize('solver', ['auto', 'bicgstab'])
def test_plume_fronts_match_the_direct_solve_element_by_element(gpu_lib, path, nx, ny, solver):
    """Constituents 3 and 7 of distinct_input_array are EXACTLY zero outside a disc: the implicit solution then
    decays through every decade down to 1e-300 ahead of the front, and a 2-norm stopping rule cannot see those
    cells.  rel_err asserts |a-b| <= 1e-6 |b| + 1e-12 max|b| for every cell, level and constituent."""
    import clearwater_riverine_amd as cw
    K, steps = 8, 4
    mesh, inputs3 = distinct_case(K, nx=nx, ny=ny, n_steps=steps, seed=3, n_merge=nx * ny // 20, dt=40.0,
                                  diffusion_coefficient=0.5)
    names = [f'c{k}' for k in range(K)]
    ref = oracle_run(mesh, inputs3, steps)
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)},
                                  solver=solver)
    for _ in range(steps):
        model.update()
        assert model.last_step.flags == 0
    assert model.last_step.sweep_kernel == ((7 if nx == 70 else 6) if solver == 'auto' else 0)
    n = mesh['nreal'] + 1
    plume = ref.constituent_dict['c3'].state[steps, :n]
    if nx == 200:                                          # (the short mesh is filled by diffusion within four steps)
        assert np.count_nonzero((plume > 0) & (plume < 1e-9 * plume.max())) > 100      # the fronts are really there
    for nm in names:
        assert rel_err(model.mesh[nm], ref.constituent_dict[nm].state) <= 1e-9
        assert flux_err(model.constituent_dict[nm].total_mass_flux[:steps], ref.constituent_dict[nm].total_mass_flux[:steps]) <= 1e-8


def test_element_wise_rule_is_what_holds_the_fronts(gpu_lib, monkeypatch):
    """A/B of the rule itself: with it switched off (CWR_NO_ELEMENTWISE=1) the same run still satisfies the norm
    criterion but needs fewer sweeps -- and with it on the sweeps rise, i.e. the rule is the binding one here."""
    import clearwater_riverine_amd as cw
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    mesh, inputs3 = distinct_case(4, nx=200, ny=40, n_steps=2, seed=3, n_merge=400, dt=40.0, diffusion_coefficient=0.5)
    n = mesh['nreal'] + 1
    sweeps = {}
    for off in ('1', '0'):
        monkeypatch.setenv('CWR_NO_ELEMENTWISE', off)
        eng = make_engine(mesh, inputs3)
        eng.set_state(inputs3[0, :n, :])
        r = [eng.step(t, tol=1e-12) for t in range(2)]
        assert all(x.max_rel_residual <= 1e-12 for x in r)
        sweeps[off] = r[-1].sweeps
        eng.close()
    assert sweeps['0'] >= sweeps['1']


@pytest.mark.parametrize('nx,ny', [(30, 12), (120, 50)])
@pytest.mark.parametrize('solver', ['auto', 'jacobi', 'bicgstab'])
def test_a_failed_step_leaves_the_state_untouched_and_can_be_retried(gpu_lib, nx, ny, solver):
    """The solver iterates in place in the state vector; a step that fails (here: max_iter = 3) must put x_t and the
    ghost rows back, so that a caller who catches SolverNotConverged and calls update() again gets the reference's
    result -- through the facade, whose _device_level still says `t`."""
    import clearwater_riverine_amd as cw
    K = 3
    mesh, inputs3 = distinct_case(K, nx=nx, ny=ny, n_steps=4, seed=5, n_merge=20, dt=60.0, diffusion_coefficient=0.3)
    names = [f'c{k}' for k in range(K)]
    ref = oracle_run(mesh, inputs3, 4)
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)},
                                  solver=solver)
    model.update()
    model.update()
    before = model.engine.get_state()
    model.max_iter = 3
    with pytest.raises(cw.SolverNotConverged):
        model.update()
    assert model.time_step == 2
    assert np.array_equal(model.engine.get_state(), before, equal_nan=True)      # bitwise: real cells AND ghost rows
    model.max_iter = 5000
    model.update()                                                               # the retry
    model.update()
    for nm in names:
        assert rel_err(model.mesh[nm], ref.constituent_dict[nm].state) <= 1e-9


def test_nan_failure_restores_the_state_too(gpu_lib):
    import clearwater_riverine_amd as cw
    mesh, inputs3 = distinct_case(2, nx=40, ny=16, n_steps=2, seed=6, n_merge=10)
    n = mesh['nreal'] + 1
    eng = make_engine(mesh, inputs3)
    x = inputs3[0, :n, :].copy()
    x[5, 1] = np.nan
    eng.set_state(x)
    with pytest.raises(FloatingPointError):
        eng.step(0)
    assert np.array_equal(eng.get_state()[:n], x, equal_nan=True)
    eng.set_state(inputs3[0, :n, :])
    eng.step(0)                                                                  # and the engine is still usable


def test_stiff_small_mesh_hands_over_to_bicgstab_in_auto_mode(gpu_lib):
    """A mesh that fits the one-launch LDS solver, stiff enough (dt = 20 000 s on 10 m cells) that the sweeps run out of
    a small budget: in 'auto' mode BiCGSTAB must continue from the iterate -- scipy's spsolve simply succeeds there --
    while 'jacobi' (forced) reports SolverNotConverged."""
    import clearwater_riverine_amd as cw
    mesh = cw.synthetic.make_mesh(24, 10, 3, seed=3, dt=20000.0, breathing=0.0, n_merge=6)
    oracle.derive_coefficients(mesh)
    inputs3 = cw.synthetic.distinct_input_array(mesh, 2, seed=1)
    n = mesh['nreal'] + 1
    eng = make_engine(mesh, inputs3)
    eng.set_state(inputs3[0, :n, :])
    with pytest.raises(cw.SolverNotConverged):
        eng.step(0, max_iter=40, solver='jacobi')
    # ||J||_inf > 0.9967 at this dt, which used to clamp the scale of the element-wise rule (with a warning).  Round 4: the
    # row-wise bound max((I - J)^-1 1) - 1 is what scales it, and on a mesh 24 cells long that is a few dozen, not 1 / (1 - ||J||_inf)
    assert eng.jacobi_norms()[0] > 0.9967 and eng.error_factors()[0] < 100.0
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        res = eng.step(0, max_iter=40, solver='auto')
    assert res.sweep_kernel == 7 and res.solver == 2 and res.iterations > 0 and res.flags == 0
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(2)})
    ref.update()
    got = eng.get_state()
    for k in range(2):
        assert rel_err(got[:, k], ref.constituent_dict[f'c{k}'].state[1]) <= 1e-9


@pytest.mark.parametrize('nx,ny', [(30, 12), (110, 50)])
def test_real_cell_inputs_at_later_levels_follow_the_reference(gpu_lib, nx, ny):
    """input_array rows t >= 1 with non-zero entries on REAL cells (a point source held at a fixed concentration):
    the reference writes them into the solved level before _mass_flux (transport.py:258-264) and starts the next step
    from them (linalg.py:199-200).  State AND fluxes of every level must match the oracle."""
    import clearwater_riverine_amd as cw
    K, steps = 3, 5
    mesh, inputs3 = distinct_case(K, nx=nx, ny=ny, n_steps=steps, seed=8, n_merge=15, dt=30.0, diffusion_coefficient=0.2)
    n = mesh['nreal'] + 1
    rng = np.random.default_rng(0)
    src = rng.choice(n, size=7, replace=False)
    inputs3[1:4, src[:4], 0] = 250.0                      # constituent 0: four cells held for levels 1-3
    inputs3[2, src[3:], 2] = 40.0 + rng.random(4)         # constituent 2: one level, overlapping cell src[3]
    names = [f'c{k}' for k in range(K)]
    ref = oracle_run(mesh, inputs3, steps)
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)})
    for _ in range(steps):
        model.update()
    assert model.mesh['c0'][2, src[0]] == 250.0
    for nm in names:
        assert rel_err(model.mesh[nm], ref.constituent_dict[nm].state) <= 1e-9
        for got, want in ((model.constituent_dict[nm].advection_mass_flux, ref.constituent_dict[nm].advection_mass_flux),
                          (model.constituent_dict[nm].total_mass_flux, ref.constituent_dict[nm].total_mass_flux)):
            assert flux_err(got[:steps], want[:steps]) <= 1e-8


def _oracle_jacobi_norm(mesh, t):
    """||J||_inf of step t from the oracle's literal matrix: max over rows of sum_{j != i} |A_ij| / A_ii."""
    lhs = oracle.LHS(mesh)
    lhs.update_values(mesh, t)
    n = mesh['nreal'] + 1
    A = lhs.csr().tocsr()[:n, :n]
    d = A.diagonal()
    off = abs(A).sum(axis=1).A1 - np.abs(d)
    return float(np.max(off / d))


def test_jacobi_norms_of_the_loaded_flow_field_match_the_oracle_matrix(gpu_lib):
    """cwr_get_jacobi_norms (k_jnorm, evaluated when the flow field is loaded) against the duplicate-bearing COO matrix of
    linalg.py:34-156 assembled by the oracle: the number that scales the element-wise rule is the exact max-norm
    contraction of the Jacobi iteration, not a measured rate."""
    import clearwater_riverine_amd as cw
    mesh, inputs3 = distinct_case(2, nx=60, ny=24, n_steps=5, seed=4, n_merge=70, n_dry=2, dt=120.0, diffusion_coefficient=0.4)
    eng = make_engine(mesh, inputs3)
    got = eng.jacobi_norms()
    assert got.shape == (6,) and got[-1] == 0.0
    wants = [_oracle_jacobi_norm(mesh, t) for t in range(5)]
    for t in range(5):
        assert got[t] == pytest.approx(wants[t], rel=1e-12)
    # (the two dry cells of this mesh break continuity for their neighbours -- inflow without the matching outflow -- so some
    # rows are NOT diagonally dominant and ||J||_inf > 1: the NORM form of the bound does not exist there.  Round 4: the scale of
    # the element-wise rule is the row-wise bound max((I - J)^-1 1) - 1, which does: checked against the oracle's matrix, and the
    # step runs without the clamp flag)
    assert max(wants) > 1.0
    n = mesh['nreal'] + 1
    F = eng.error_factors()
    for t in range(5):
        true_F = _oracle_error_factor(mesh, t)
        assert true_F <= F[t] * (1 + 1e-12) and F[t] <= max(3.0, 1.15 * true_F) + 1e-9, (t, true_F, F[t])      # a bound, and a tight one (sweeps stop below 3)
    assert F.max() < 100.0
    eng.set_state(inputs3[0, :n, :])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        assert eng.step(0).flags == 0
    # a caller's own norms bring the norm form back (and with it the clamp)
    eng.set_jacobi_norms(eng.jacobi_norms())
    assert not np.isfinite(eng.error_factors()[:5]).all()
    with pytest.warns(RuntimeWarning, match='element-wise'):
        assert eng.step(1).flags == cw.engine.INFO_ELEMENTWISE_CLAMPED
    eng.close()


def _oracle_error_factor(mesh, t):
    """max((I - J)^-1 1) - 1 of step t from the oracle's literal matrix: (D^-1 A) w = 1 by spsolve."""
    import scipy.sparse as sp
    from scipy.sparse.linalg import spsolve
    lhs = oracle.LHS(mesh)
    lhs.update_values(mesh, t)
    n = mesh['nreal'] + 1
    A = lhs.csr().tocsr()[:n, :n]
    w = spsolve((sp.diags(1.0 / A.diagonal()) @ A).tocsc(), np.ones(n))
    assert (w >= 1.0 - 1e-12).all()
    return float(w.max() - 1.0)


@pytest.mark.parametrize('K,nx,ny', [(3, 90, 40), (16, 64, 30)])
def test_a_third_of_the_cells_dry_matches_the_oracle_element_wise_without_flags(gpu_lib, K, nx, ny, monkeypatch):
    """HEC-RAS floodplain meshes are mostly dry most of the time.  30 % permanently dry cells (volume 0, no
    flow on their faces: linalg.py:76-81 gives them a dummy diagonal) at CFL ~ 2: the wet neighbours of dry cells have row sums
    above 1, so ||J||_inf says nothing -- the row-wise bound keeps the element-wise rule inside its working range: no
    ELEMENTWISE_CLAMPED, and every wet cell within 1e-6 of its own spsolve value."""
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    steps = 4
    mesh, inputs3 = distinct_case(K, nx=nx, ny=ny, n_steps=steps, seed=12, n_merge=nx * ny // 25, n_dry=int(0.3 * nx * ny), dt=30.0,
                                  diffusion_coefficient=0.5)
    n = mesh['nreal'] + 1
    assert np.count_nonzero(np.asarray(mesh['volume'])[0, :n] == 0) >= 0.29 * n
    ref = oracle_run(mesh, inputs3, steps)
    eng = make_engine(mesh, inputs3)
    assert eng.jacobi_norms()[:steps].max() > 1.0
    assert eng.error_factors()[:steps].max() < 300.0
    eng.set_state(inputs3[0, :n, :])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        for t in range(steps):
            assert eng.step(t, tol=1e-12, max_iter=20000).flags == 0
    got = eng.get_state()
    for k in range(K):
        assert rel_err(got[:n, k], ref.constituent_dict[f'c{k}'].state[steps, :n]) <= 1e-9
    eng.close()


@pytest.mark.parametrize('dt,split,expect_clamp', [(2000.0, 1, False), (20000.0, 1, False), (20000.0, 0, True)])
def test_plume_fronts_in_the_stiff_regime_and_the_clamp_flag(gpu_lib, dt, split, expect_clamp, monkeypatch):
    """the element-wise rule at CFL >= 100.  dt = 2000 s on 10 m cells at 0.5 m/s is CFL 100 (||J||_inf ~ 0.99):
    the scale s = 0.3 (1 - rho) / rho is above 1e-3, no flag, and every cell of the plume constituents -- fronts many decades below the
    peak included -- is within 1e-6 of ITS OWN spsolve value.  dt = 20 000 s (CFL 1000, ||J||_inf > 0.9967, F > 300): since round 6
    only the absolute part of the rule is floored at s = 1e-3, the relative part follows 0.3 / F -- no flag, no warning, and the FULL
    element-wise bar against spsolve (until then the step ran at s = 1e-3, said so, and was held to 1e-4 only).
    CWR_EW_SPLIT=0: round 5's rule -- the step says CWR_INFO_ELEMENTWISE_CLAMPED (-> one RuntimeWarning)."""
    import warnings
    import clearwater_riverine_amd as cw
    monkeypatch.setenv('CWR_EW_SPLIT', str(split))
    K, steps = 4, 2
    mesh, inputs3 = distinct_case(K, nx=200, ny=40, n_steps=steps, seed=3, n_merge=400, dt=dt, diffusion_coefficient=0.5,
                                  breathing=0.0)
    names = [f'c{k}' for k in range(K)]
    ref = oracle_run(mesh, inputs3, steps)
    n = mesh['nreal'] + 1
    eng = make_engine(mesh, inputs3)
    rho = eng.jacobi_norms()[:steps]
    assert (rho > 0.9967).all() if dt > 2000.0 else ((rho > 0.98).all() and (rho < 0.9967).all())
    assert ((eng.error_factors()[:steps] > 300.0).all() and np.isfinite(eng.error_factors()[:steps]).all()) if dt > 2000.0 else (eng.error_factors()[:steps] < 300.0).all()
    eng.set_state(inputs3[0, :n, :])
    flags = []
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter('always')
        for t in range(steps):
            flags.append(eng.step(t, tol=1e-12, max_iter=200000).flags)
    clamped = [bool(f & cw.engine.INFO_ELEMENTWISE_CLAMPED) for f in flags]
    assert all(clamped) if expect_clamp else not any(flags)
    assert bool(seen) == expect_clamp                     # the wrapper never swallows a tolerance decision
    got = eng.get_state()
    for k, nm in enumerate(names):
        want = ref.constituent_dict[nm].state[steps]
        if expect_clamp:                                  # the rigorous part of round 5's rule: max-norm forward error
            assert rel_err(got[:, k], want, ew_rtol=1e-4, ew_atol=1e-9) <= 1e-6
        else:
            assert rel_err(got[:, k], want) <= 1e-9       # element-wise 1e-6 |b| + 1e-12 max|b|
    eng.close()


def test_tolerance_changed_between_steps_of_one_engine(gpu_lib, monkeypatch):
    """the batch graphs are captured once and replayed by later steps; the relative element-wise tolerance used to
    be a by-value kernel argument frozen into them.  Loose steps (tol = 1e-8) capture the graphs, then the same engine is
    asked for tol = 1e-12 from an exact state: the fronts must meet the element-wise bar against spsolve."""
    import clearwater_riverine_amd as cw
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    K, steps = 4, 5
    mesh, inputs3 = distinct_case(K, nx=200, ny=40, n_steps=steps, seed=3, n_merge=400, dt=40.0, diffusion_coefficient=0.5)
    n = mesh['nreal'] + 1
    ref = oracle_run(mesh, inputs3, steps)
    eng = make_engine(mesh, inputs3)
    eng.set_state(inputs3[0, :n, :])
    for t in range(steps - 1):
        assert eng.step(t, tol=1e-8).max_rel_residual <= 1e-8
    exact = np.stack([ref.constituent_dict[f'c{k}'].state[steps - 1, :n] for k in range(K)], axis=1)
    eng.set_state(exact)
    r = eng.step(steps - 1, tol=1e-12)
    assert r.max_rel_residual <= 1e-12 and r.flags == 0
    got = eng.get_state()
    for k in range(K):
        assert rel_err(got[:, k], ref.constituent_dict[f'c{k}'].state[steps]) <= 1e-9
    eng.close()
