"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the CPU oracle.

Tolerances: indices bit-exact by construction (same ids at the boundary); the operator and the
right-hand side within 1e-12 relative (same arithmetic, different summation order); concentrations
after implicit steps within 1e-9 relative (north star: <= 1e-6) of the oracle's SuperLU solution.
"""
import numpy as np
import pytest

import cwr_oracle as oracle
from util import flux_err, load_plan, multi_inputs, oracle_run, rel_err

pytestmark = pytest.mark.gpu

TOL_OP = 1e-12
TOL_CONC = 1e-9


def make_engine(mesh, inputs3):
    import clearwater_riverine_amd as cw
    n = mesh['nreal'] + 1
    ncell = len(mesh['face_x'])
    K = inputs3.shape[2]
    eng = cw.TransportEngine(mesh['edges_face1'], mesh['edges_face2'], ncell, K)
    eng.load_flow_field(mesh['face_flow'], mesh['edge_velocity'], mesh['volume'], mesh['dt'],
                        mesh['face_to_face_dist'], mesh['diffusion_coefficient'])
    eng.load_boundary(inputs3[:, n:, :])
    return eng


def synthetic_case(K, **kw):
    import clearwater_riverine_amd as cw
    mesh = cw.synthetic.make_mesh(**kw)
    oracle.derive_coefficients(mesh)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K)
    return mesh, inputs3


@pytest.mark.parametrize('plan,D', [('plan01', 0.01), ('plan02', 0.01), ('plan03', 0.001), ('plan01', 0.0)])
def test_device_coefficients_match_reference_derivation(gpu_lib, plan, D):
    """a-1 on device == utilities.py:513-535 restated (float32 products, float64 division)."""
    mesh, inp, _ = load_plan(plan, D)
    eng = make_engine(mesh, inp[:, :, None])
    for t in (0, 1, len(mesh['dt']) - 1):
        adv, dif = eng.get_coefficients(t)
        assert np.array_equal(adv, mesh['advection_coeff'][t])            # bit-exact float32
        assert np.array_equal(dif, mesh['coeff_to_diffusion'][t])         # bit-exact float64


@pytest.mark.parametrize('plan,D', [('plan01', 0.01), ('plan02', 0.01), ('plan03', 0.001)])
@pytest.mark.parametrize('K', [1, 2, 3, 16])
def test_apply_matches_coo_assembly(gpu_lib, plan, D, K):
    """y = A x against the entry-by-entry COO assembly + csr_matrix of linalg.py:34-156."""
    mesh, inp, _ = load_plan(plan, D)
    eng = make_engine(mesh, multi_inputs(inp, K))
    n = mesh['nreal'] + 1
    lhs = oracle.LHS(mesh)
    rng = np.random.default_rng(7)
    for t in (0, 1, 5, 20):
        lhs.update_values(mesh, t)
        A = lhs.csr()
        x = rng.standard_normal((n, K))
        y = eng.apply(t, x)
        ref = A @ x
        assert np.max(np.abs(y - ref)) <= TOL_OP * np.max(np.abs(ref))


@pytest.mark.parametrize('K', [1, 2, 5, 12, 16])
def test_apply_synthetic_mixed_degree(gpu_lib, K):
    mesh, inputs3 = synthetic_case(K, nx=40, ny=17, n_steps=6, seed=3, n_merge=60, n_dry=4)
    eng = make_engine(mesh, inputs3)
    n = mesh['nreal'] + 1
    lhs = oracle.LHS(mesh)
    rng = np.random.default_rng(1)
    for t in (0, 3, 5):
        lhs.update_values(mesh, t)
        A = lhs.csr()
        x = rng.standard_normal((n, K))
        ref = A @ x
        assert np.max(np.abs(eng.apply(t, x) - ref)) <= TOL_OP * np.max(np.abs(ref))


@pytest.mark.parametrize('K', [1, 2, 3, 16])
def test_rhs_matches_reference_assembly(gpu_lib, K):
    """b against RHS.update_values restated literally (incl. last-write-wins at corner cells)."""
    mesh, inputs3 = synthetic_case(K, nx=12, ny=7, n_steps=6, seed=5, n_merge=6)
    eng = make_engine(mesh, inputs3)
    n = mesh['nreal'] + 1
    rng = np.random.default_rng(2)
    for t in (0, 2, 4):
        x = 1.0 + rng.random((n, K))
        b = eng.rhs(t, x)
        for k in range(K):
            r = oracle.RHS(mesh, inputs3[:, :, k].copy())
            # t > 0: input_array[t] has no real-cell entries, so the solution passes through
            r.input_array[t, :n] = 0.0
            r.update_values(x[:, k], mesh, t)
            assert np.max(np.abs(b[:, k] - r.vals)) <= TOL_OP * np.max(np.abs(r.vals))


@pytest.mark.parametrize('solver', ['auto', 'bicgstab', 'multi-launch'])
@pytest.mark.parametrize('plan,D,steps', [('plan01', 0.01, 30), ('plan02', 0.01, 24), ('plan03', 0.001, 30),
                                          ('plan01', 0.0, 10)])
def test_facade_update_matches_oracle_on_reference_fixtures(gpu_lib, plan, D, steps, solver, monkeypatch):
    """ClearwaterRiverine.update() loop == oracle (spsolve) on the reference's HDF fixtures:
    concentrations incl. ghost-cell NaN pattern, and the three mass-flux arrays."""
    import clearwater_riverine_amd as cw
    if solver == 'multi-launch':
        monkeypatch.setenv('CWR_NO_SMALL', '1')
        solver = 'auto'
    mesh, inp, _ = load_plan(plan, D)
    ref = oracle_run(mesh, inp[:, :, None], steps)
    model = cw.ClearwaterRiverine(mesh=dict(mesh), diffusion_coefficient_input=D, input_arrays={'c0': inp.copy()},
                                  solver=solver)
    for _ in range(steps):
        model.update()
    assert model.time_step == steps
    rc = ref.constituent_dict['c0']
    assert rel_err(model.mesh['c0'][:steps + 1], rc.state[:steps + 1]) <= TOL_CONC
    mc = model.constituent_dict['c0']
    for got, want in ((mc.advection_mass_flux, rc.advection_mass_flux), (mc.diffusion_mass_flux, rc.diffusion_mass_flux),
                      (mc.total_mass_flux, rc.total_mass_flux)):
        assert flux_err(got[:steps], want[:steps]) <= 1e-8


@pytest.mark.parametrize('path', ['one-launch small-mesh solver', 'multi-launch sweeps'])
@pytest.mark.parametrize('solver', ['auto', 'jacobi', 'bicgstab'])
@pytest.mark.parametrize('K', [1, 3, 12, 16])
def test_facade_multi_constituent_and_override(gpu_lib, K, solver, path, monkeypatch):
    """K batched constituents + the update_concentration override of transport.py:233-236, through both Jacobi
    paths (the LDS-resident one-launch solver of small meshes, and the tiled sweeps / J^2 passes of large ones)."""
    import clearwater_riverine_amd as cw
    if path == 'multi-launch sweeps':
        monkeypatch.setenv('CWR_NO_SMALL', '1')
    mesh, inputs3 = synthetic_case(K, nx=30, ny=11, n_steps=12, seed=11, n_merge=25, n_dry=2)
    names = [f'c{k}' for k in range(K)]
    n = mesh['nreal'] + 1
    rng = np.random.default_rng(4)
    overrides = {5: {names[0]: 2.0 + rng.random(n)}, 8: {names[-1]: 3.0 + rng.random(n), 'not_a_constituent': np.zeros(n)}}
    ref = oracle_run(mesh, inputs3, 12, overrides)
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)},
                                  solver=solver)
    for s in range(12):
        model.update(overrides.get(s))
    for nm in names:
        assert rel_err(model.mesh[nm], ref.constituent_dict[nm].state) <= TOL_CONC
        assert flux_err(model.constituent_dict[nm].total_mass_flux[:12], ref.constituent_dict[nm].total_mass_flux[:12]) <= 1e-8


@pytest.mark.parametrize('K,pad', [(2, True), (3, True), (5, True), (5, False), (7, True), (7, False), (8, True), (10, True), (10, False), (12, True),
                                   (13, True), (20, True), (24, True), (32, True), (64, True)])
def test_step_parity_across_constituent_counts_on_tiled_meshes(gpu_lib, K, pad, monkeypatch):
    """Every lane mapping of the sweep kernels (VW = 1 for odd K, 2, and the four-wide split-row mapping of
    K % 16 == 0; wide rows that do not fit the tiled pass fall back to the un-tiled J^2 pass) against the oracle's
    spsolve, on a mesh of many 64-row tiles with merged (5-6 face) cells and a dry cell.
    Round 5: K = 3, 5, 7, 10, 13 run as 4, 6, 8, 12, 16 inside the engine (zero columns behind the caller's, stripped at every
    read-out: cwr_create); pad = False (CWR_K_PAD=0) keeps the native odd / 5-lane mappings exercised."""
    import clearwater_riverine_amd as cw
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    if not pad:
        monkeypatch.setenv('CWR_K_PAD', '0')
    mesh, inputs3 = synthetic_case(K, nx=96, ny=48, n_steps=3, seed=13, n_merge=150, n_dry=1, dt=30.0,
                                   diffusion_coefficient=0.4)
    n = mesh['nreal'] + 1
    names = [f'c{k}' for k in range(K)]
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)},
                                  store_history=True)
    ref = oracle_run(mesh, inputs3[:, :, [0, K - 1]], 3)        # the oracle solves per constituent: first and last suffice
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')                            # (round 4: the dry cell no longer voids the error bound -- no clamp warning)
        for _ in range(3):
            model.update()
            assert model.last_step.sweep_kernel in (5, 6) and model.last_step.max_rel_residual <= 1e-12
            assert model.last_step.flags == 0
    for kk, nm in ((0, names[0]), (1, names[-1])):
        assert rel_err(model.mesh[nm], ref.constituent_dict[f'c{kk}'].state) <= TOL_CONC
        assert flux_err(model.constituent_dict[nm].total_mass_flux[:3], ref.constituent_dict[f'c{kk}'].total_mass_flux[:3]) <= 1e-8
    want_stride = {3: 4, 5: 6, 7: 8, 10: 12, 13: 16}.get(K, K) if pad else K
    assert model.engine.state_row_stride() == want_stride
    # the blocking read-outs strip the padded columns too (the facade above read through the output ring)
    st = model.engine.get_state()
    assert st.shape == (len(mesh['face_x']), K) and rel_err(st[:, K - 1], ref.constituent_dict['c1'].state[3]) <= TOL_CONC
    adv, dif, tot = model.engine.get_mass_flux()
    assert tot.shape == (len(mesh['edges_face1']), K) and flux_err(tot[:, 0], ref.constituent_dict['c0'].total_mass_flux[2]) <= 1e-8
    mass, vol = model.engine.domain_mass(3)
    assert mass.shape == (K,) and np.isclose(mass[K - 1], float(np.sum(np.asarray(mesh['volume'])[3, :n].astype(np.float64) * st[:n, K - 1])), rtol=1e-9)


@pytest.mark.mid_mesh_default
@pytest.mark.parametrize('K', [1, 12])
@pytest.mark.parametrize('n_target,path,dry', [(2943, 'default', 0), (10000, 'default', 0), (10000, 'tiled passes', 0),
                                               (9750, 'default', 1), (9750, 'tiled passes', 1)])
def test_baseline_configs_2_and_3_river_band_mesh(gpu_lib, K, n_target, path, dry, monkeypatch):
    """BASELINE configs 2 / 3: a river-band mesh of the Ohio River's size (2 943 cells; 10 000 nominal) with
    jittered, partly merged 5-6-sided cells and a locally shuffled numbering, dt = 3600 s (CFL ~ 18), one tracer and the
    12-constituent NSM-I state vector, through the facade against the oracle's spsolve.  2 943 cells take the one-launch
    LDS-resident solver with one workgroup per constituent, 10 000 the same with several (round 5) -- or, with
    CWR_SMALL_MAX_CELLS=0, the tiled block-asynchronous passes as before.
    dry (round 6): the reference's everyday regime as tools' matrix probe builds it -- 200 x 50 = 9 750 cells of
    75 m, 0.5 % DRY cells (linalg.py:66,76-81: dummy diagonal 1), breathing volumes, dt = 3600 s -- 24 steps, every level element-wise
    against spsolve, with flags == 0 and no warning on any step (until round 5 such a step ran with CWR_INFO_ELEMENTWISE_CLAMPED:
    F = 185 ... 323 > 300; the rule's relative part is no longer floored there)."""
    if path == 'tiled passes':
        monkeypatch.setenv('CWR_SMALL_MAX_CELLS', '0')
    import warnings
    import clearwater_riverine_amd as cw
    steps = 24 if dry else 6
    if dry:
        mesh = cw.synthetic.make_mesh(200, 50, steps, seed=20100529, n_merge=200 * 50 // 40, dx=75.0, dy=75.0, depth=3.0, dt=3600.0, velocity=0.3,
                                      breathing=0.1, diffusion_coefficient=0.1, period_steps=24, n_dry=200 * 50 // 200)
        assert int((np.asarray(mesh['volume'])[1, :mesh['nreal'] + 1] == 0).sum()) >= 40
    else:
        nx, ny, nm = (109, 28, 109) if n_target == 2943 else (200, 51, 200)
        mesh = cw.synthetic.make_mesh(nx, ny, steps, seed=20100529 % 100000, n_merge=nm, dx=75.0, dy=75.0, dt=3600.0, velocity=0.4,
                                      diffusion_coefficient=0.1, period_steps=24)
    oracle.derive_coefficients(mesh)
    n = mesh['nreal'] + 1
    assert abs(n - n_target) <= 0.01 * n_target
    inputs3 = cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=24 * 3600.0)
    names = [f'c{k}' for k in range(K)]
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm_: inputs3[:, :, k].copy() for k, nm_ in enumerate(names)})
    cols = [0, K - 1] if K > 1 else [0]
    ref = oracle_run(mesh, inputs3[:, :, cols], steps)
    with warnings.catch_warnings():
        warnings.simplefilter('error')                   # (no tolerance decision may be announced: RuntimeWarning -> failure)
        for _ in range(steps):
            model.update()
            assert model.last_step.flags == 0 and model.last_step.iterations == 0
    assert model.last_step.sweep_kernel == (6 if path == 'tiled passes' else 7)
    if dry:
        F = model.engine.error_factors()[:steps]
        assert F.max() > 150.0 and np.isfinite(F).all()   # (the regime the test is for: at or beyond round 5's clamp threshold of 300 -- 185 ... 323 over 30 levels, profiles/r06_matrix_probe.txt)
    for kk, col in enumerate(cols):
        assert rel_err(model.mesh[names[col]], ref.constituent_dict[f'c{kk}'].state) <= TOL_CONC
        assert flux_err(model.constituent_dict[names[col]].total_mass_flux[:steps], ref.constituent_dict[f'c{kk}'].total_mass_flux[:steps]) <= 1e-8


@pytest.mark.parametrize('K,steps', [(1, 912), (12, 96)])
def test_baseline_config_2_full_length_run_on_the_ohio_sized_mesh(gpu_lib, K, steps):
    """BASELINE config 2 at its specified length: 912 hourly steps of one tracer on the 2 943-cell river-band
    mesh (the Ohio River model's size; its HDF is a missing blob), and 96 steps of the 12-constituent NSM-I state vector
    (config 3), step by step against the oracle's spsolve: every stored level, element-wise."""
    import clearwater_riverine_amd as cw
    mesh = cw.synthetic.make_mesh(109, 28, steps, seed=20100529 % 100000, n_merge=109, dx=75.0, dy=75.0, dt=3600.0, velocity=0.4,
                                  diffusion_coefficient=0.1, period_steps=24)
    oracle.derive_coefficients(mesh)
    n = mesh['nreal'] + 1
    assert abs(n - 2943) <= 30
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=7)
    names = [f'c{k}' for k in range(K)]
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm_: inputs3[:, :, k].copy() for k, nm_ in enumerate(names)})
    ref = oracle_run(mesh, inputs3, steps)
    for _ in range(steps):
        model.update()
        assert model.last_step.flags == 0
    assert model.time_step == steps and model.last_step.sweep_kernel == 7
    for k, nm in enumerate(names):
        assert rel_err(model.mesh[nm], ref.constituent_dict[f'c{k}'].state) <= TOL_CONC
        assert flux_err(model.constituent_dict[nm].total_mass_flux[:steps], ref.constituent_dict[f'c{k}'].total_mass_flux[:steps]) <= 1e-8


@pytest.mark.parametrize('K', [1, 16])
def test_dense_adjacency_mesh_is_tiled_and_matches_the_oracle(gpu_lib, K, monkeypatch):
    """A third of the cells merged into 5-6-face cells (4.7 faces and 12 J^2 entries per row on average): a 256-row tile
    of a one-constituent engine then holds more entries than any compiled configuration, and the engine must retry with
    half-size tiles instead of falling back to the un-tiled exact pass."""
    import clearwater_riverine_amd as cw
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    mesh, inputs3 = synthetic_case(K, nx=120, ny=60, n_steps=3, seed=7, n_merge=1800, dt=60.0, diffusion_coefficient=0.3)
    n = mesh['nreal'] + 1
    names = [f'c{k}' for k in range(K)]
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)})
    cols = [0, K - 1] if K > 1 else [0]
    ref = oracle_run(mesh, inputs3[:, :, cols], 3)
    for _ in range(3):
        model.update()
        assert model.last_step.sweep_kernel == 6 and model.last_step.max_rel_residual <= 1e-12
    for kk, col in enumerate(cols):
        assert rel_err(model.mesh[names[col]], ref.constituent_dict[f'c{kk}'].state) <= TOL_CONC


@pytest.mark.parametrize('K,small', [(1, False), (4, False), (16, False), (3, True)])
def test_meshes_with_eight_sided_cells_match_the_oracle(gpu_lib, K, small, monkeypatch):
    """HEC-RAS cells have up to 8 faces.  2 x 2 blocks of quads merged into 8-sided cells (two faces towards each
    neighbour pair, J^2 rows of up to ~24 entries) beside 6-sided and plain cells: the 8-face instantiations of the set-up
    and sweep kernels (k_sq_numeric<8> with its whole-row vector gathers, the tiled pass's long rows, k_small_jacobi's
    8-face records) against the oracle's spsolve, fluxes included."""
    import clearwater_riverine_amd as cw
    if small:
        kw = dict(nx=40, ny=30, n_steps=3, seed=31, n_merge=60, n_merge4=50, dt=30.0, diffusion_coefficient=0.2)
    else:
        monkeypatch.setenv('CWR_NO_SMALL', '1')
        kw = dict(nx=120, ny=64, n_steps=3, seed=31, n_merge=500, n_merge4=400, dt=40.0, diffusion_coefficient=0.3)
    mesh, inputs3 = synthetic_case(K, **kw)
    n = mesh['nreal'] + 1
    deg = np.bincount(np.concatenate([mesh['edges_face1'], mesh['edges_face2']]), minlength=len(mesh['face_x']))[:n]
    assert deg.max() == 8
    names = [f'c{k}' for k in range(K)]
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)})
    cols = [0, K - 1] if K > 1 else [0]
    ref = oracle_run(mesh, inputs3[:, :, cols], 3)
    for _ in range(3):
        model.update()
        assert model.last_step.max_rel_residual <= 1e-12
        assert model.last_step.sweep_kernel == (7 if small else 6)
    for kk, col in enumerate(cols):
        assert rel_err(model.mesh[names[col]], ref.constituent_dict[f'c{kk}'].state) <= TOL_CONC
        assert flux_err(model.constituent_dict[names[col]].total_mass_flux[:3], ref.constituent_dict[f'c{kk}'].total_mass_flux[:3]) <= 1e-8


@pytest.mark.parametrize('nx,ny,rpt', [(25, 20, 1), (50, 38, 2), (72, 42, 3), (90, 45, 4)])
def test_one_launch_solver_at_every_rows_per_thread_variant(gpu_lib, nx, ny, rpt):
    """k_small_jacobi deals the rows to its 1 024 threads sorted by neighbour count, 1 / 2 / 3 / 4 rows per thread (round 5): meshes
    of each size class with 8-sided, 6-sided, plain and dry cells, K = 3 (carried as 4), against the oracle's spsolve -- states,
    fluxes, and the same answer as the multi-launch path to solver tolerance."""
    import clearwater_riverine_amd as cw
    K, steps = 3, 4
    mesh, inputs3 = synthetic_case(K, nx=nx, ny=ny, n_steps=steps, seed=17 + rpt, n_merge=nx * ny // 30, n_merge4=nx * ny // 60, n_dry=3,
                                   dt=30.0, diffusion_coefficient=0.2)
    n = mesh['nreal'] + 1
    assert (n + 1023) // 1024 == rpt
    ref = oracle_run(mesh, inputs3, steps)
    names = [f'c{k}' for k in range(K)]
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)})
    for _ in range(steps):
        model.update()
        assert model.last_step.sweep_kernel == 7 and model.last_step.max_rel_residual <= 1e-12
    for k, nm in enumerate(names):
        assert rel_err(model.mesh[nm], ref.constituent_dict[nm].state) <= TOL_CONC
        assert flux_err(model.constituent_dict[nm].total_mass_flux[:steps], ref.constituent_dict[nm].total_mass_flux[:steps]) <= 1e-8


def _mid_case(nx, ny, K, steps, seed):
    import clearwater_riverine_amd as cw
    mesh = cw.synthetic.make_mesh(nx, ny, steps, seed=seed, n_merge=nx * ny // 30, n_merge4=nx * ny // 80, n_dry=4, dt=60.0,
                                  diffusion_coefficient=0.3)
    oracle.derive_coefficients(mesh)
    return mesh, cw.synthetic.distinct_input_array(mesh, K, seed=seed)


@pytest.mark.mid_mesh_default
@pytest.mark.parametrize('nx,ny,parts,depth', [(110, 60, 0, 12), (150, 90, 0, 12), (150, 90, 8, 3), (100, 70, 3, 1), (128, 120, 0, 5), (260, 90, 0, 12)])
def test_one_launch_solver_with_several_parts_matches_the_oracle(gpu_lib, nx, ny, parts, depth, monkeypatch):
    """k_small_jacobi<RPT, true> (round 5): 6-23 k cells with 8-sided, 6-sided, plain and dry cells, K = 3 (carried as 4): several
    workgroups per constituent with `depth` halo layers, an exchange through global memory every `depth` sweeps.  States and fluxes
    against the oracle's spsolve over several steps; run twice: the same bits."""
    import clearwater_riverine_amd as cw
    K, steps = 3, 4
    mesh, inputs3 = _mid_case(nx, ny, K, steps, seed=nx)
    n = mesh['nreal'] + 1
    assert 4096 < n <= 24576
    monkeypatch.setenv('CWR_SMALL_PARTS', str(parts)); monkeypatch.setenv('CWR_SMALL_DEPTH', str(depth))
    ref = oracle_run(mesh, inputs3, steps)
    names = [f'c{k}' for k in range(K)]
    runs = []
    for _ in range(2):
        model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)})
        for _ in range(steps):
            model.update()
            assert model.last_step.sweep_kernel == 7 and model.last_step.max_rel_residual <= 1e-12 and model.last_step.flags == 0
        runs.append({nm: np.array(model.mesh[nm]) for nm in names})
        for nm in names:
            assert rel_err(model.mesh[nm], ref.constituent_dict[nm].state) <= TOL_CONC
            assert flux_err(model.constituent_dict[nm].total_mass_flux[:steps], ref.constituent_dict[nm].total_mass_flux[:steps]) <= 1e-8
        model.close_output(); model.engine.close()
    for nm in names:
        assert np.array_equal(runs[0][nm], runs[1][nm], equal_nan=True)


@pytest.mark.mid_mesh_default
def test_the_parts_of_the_one_launch_solver_iterate_as_one(gpu_lib, monkeypatch):
    """The halo layers are relaxed redundantly with each row's sum taken in one fixed order, so the iterates are those of the
    global Jacobi iteration whatever the number of parts: 5 / 6 / 8 parts at the same depth (= the same check cadence) stop at
    the same sweep with the same bits."""
    import clearwater_riverine_amd as cw
    K, steps = 2, 3
    mesh, inputs3 = _mid_case(140, 70, K, steps, seed=77)
    n = mesh['nreal'] + 1
    out = {}
    for parts in (5, 6, 8):
        monkeypatch.setenv('CWR_SMALL_PARTS', str(parts)); monkeypatch.setenv('CWR_SMALL_DEPTH', '6')
        eng = make_engine(mesh, inputs3)
        eng.set_state(inputs3[0, :n, :])
        sw = []
        for t in range(steps):
            r = eng.step(t, tol=1e-12)
            assert r.sweep_kernel == 7
            sw.append(r.sweeps)
        out[parts] = (sw, eng.get_state())
        eng.close()
    for parts in (6, 8):
        assert out[parts][0] == out[5][0]
        assert np.array_equal(out[parts][1], out[5][1], equal_nan=True)


@pytest.mark.mid_mesh_default
def test_meshes_without_a_plan_take_the_tiled_passes(gpu_lib, monkeypatch):
    """No admissible plan (here: at most two parts allowed for 10 k cells) -> the multi-launch path, silently, same answer."""
    import clearwater_riverine_amd as cw
    K, steps = 2, 2
    mesh, inputs3 = _mid_case(140, 70, K, steps, seed=78)
    n = mesh['nreal'] + 1
    monkeypatch.setenv('CWR_SMALL_MAX_PARTS', '2')
    eng = make_engine(mesh, inputs3)
    eng.set_state(inputs3[0, :n, :])
    ref = oracle_run(mesh, inputs3, steps)
    for t in range(steps):
        assert eng.step(t, tol=1e-12).sweep_kernel == 6
    got = eng.get_state()
    for k in range(K):
        assert rel_err(got[:n, k], ref.constituent_dict[f'c{k}'].state[steps][:n]) <= TOL_CONC
    eng.close()


@pytest.mark.mid_mesh_default
def test_the_parts_exchange_correctly_while_another_engine_loads_the_chip(gpu_lib):
    """The hand-off between the parts (sc1 stores / loads, an arrival counter) is the kind of code that passes on an idle chip and
    fails under uneven load.  Two mid-size engines step in two threads while a 190 k-cell engine keeps every CU busy from a third:
    their states must be the SAME BITS as when each ran alone (any stale halo value would change the Jacobi iterates)."""
    import threading
    import clearwater_riverine_amd as cw
    K, steps = 4, 30
    cases = [_mid_case(140, 70, K, steps, seed=91), _mid_case(128, 100, K, steps, seed=92)]
    big = cw.synthetic.make_mesh(500, 400, 6, seed=8, n_merge=10000, dt=40.0, diffusion_coefficient=0.5)
    oracle.derive_coefficients(big)
    big_in = cw.synthetic.distinct_input_array(big, 8, seed=8)

    def run(case, out, idx, stop=None):
        mesh, inputs3 = case
        n = mesh['nreal'] + 1
        eng = make_engine(mesh, inputs3)
        eng.set_state(inputs3[0, :n, :])
        sw = []
        for t in range(steps):
            r = eng.step(t, tol=1e-12)
            assert r.sweep_kernel == 7
            sw.append(r.sweeps)
        out[idx] = (sw, eng.get_state())
        eng.close()

    solo = {}
    for i, c in enumerate(cases):
        run(c, solo, i)
    stop = threading.Event()
    errors = []

    def load():
        try:
            nb = big['nreal'] + 1
            eng = make_engine(big, big_in)
            while not stop.is_set():
                eng.set_state(big_in[0, :nb, :])
                for t in range(5):
                    eng.step(t, tol=1e-12)
            eng.close()
        except Exception as exc:                      # pragma: no cover
            errors.append(exc)

    busy = {}
    loader = threading.Thread(target=load)
    loader.start()
    try:
        for rep in range(3):
            th = [threading.Thread(target=run, args=(c, busy, (rep, i))) for i, c in enumerate(cases)]
            for x in th: x.start()
            for x in th: x.join()
    finally:
        stop.set(); loader.join()
    assert not errors, errors
    bad = []
    for rep in range(3):
        for i in range(len(cases)):
            a, b = busy[(rep, i)][1], solo[i][1]
            if busy[(rep, i)][0] != solo[i][0] or not np.array_equal(a, b, equal_nan=True):
                d = np.abs(np.nan_to_num(a) - np.nan_to_num(b))
                bad.append((rep, i, busy[(rep, i)][0] == solo[i][0], float(d.max()), int((d > 0).sum()), np.nonzero(d.max(axis=0) > 0)[0].tolist()))
    assert not bad, f'(rep, engine, same sweep counts, max |difference|, entries that differ, columns that differ): {bad}'


@pytest.mark.mid_mesh_default
@pytest.mark.parametrize('note', ['note', 'download'])
def test_a_part_that_gives_up_waiting_sends_the_engine_back_to_the_tiled_passes(gpu_lib, note, monkeypatch):
    """Every wait of the one-launch solver's parts is bounded.  With the bound at zero (CWR_SMALL_SPIN_MS=0) the first part that has
    to wait at all raises the abort bit.  The parts do not agree on an abort among themselves (one may time out while the
    last arriver completes the target; the others pass, and at a final exchange write their rows): ANY part that gave up says so in
    a sticky word the host reads -- through the page-locked notification or the download (CWR_NO_NOTE=1) --, the state is restored from
    the kept copy, cwr_step solves the step with the tiled passes from the same start, and the engine stays with them: the step's flags
    carry CWR_INFO_SMALL_FALLBACK from then on (one RuntimeWarning).  Same answer as the oracle."""
    import warnings
    import clearwater_riverine_amd as cw
    K, steps = 2, 3
    mesh, inputs3 = _mid_case(140, 70, K, steps, seed=79)
    n = mesh['nreal'] + 1
    monkeypatch.setenv('CWR_SMALL_SPIN_MS', '0')
    if note == 'download':
        monkeypatch.setenv('CWR_NO_NOTE', '1')
    eng = make_engine(mesh, inputs3)
    eng.set_state(inputs3[0, :n, :])
    ref = oracle_run(mesh, inputs3, steps)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter('always')
        res = [eng.step(t, tol=1e-12) for t in range(steps)]
    kernels = [r.sweep_kernel for r in res]
    got = eng.get_state()
    eng.close()
    # (on an idle chip the parts may arrive so close together that nobody waits in the first exchanges of a step: the abort can
    # come at any step -- but once it has come, every later step takes the passes and says why)
    assert kernels == sorted(kernels, reverse=True), kernels
    assert [r.flags for r in res] == [cw.engine.INFO_SMALL_FALLBACK if kk == 6 else 0 for kk in kernels]
    said = [w for w in seen if 'multi-launch passes from here on' in str(w.message)]
    assert len(said) == (1 if 6 in kernels else 0), [str(w.message) for w in seen]
    if 6 not in kernels:                                         # pragma: no cover  (every part arrived within one poll of the last, 60 times)
        pytest.skip('no part ever had to wait: the abort path was not taken in this run')
    for k in range(K):
        assert rel_err(got[:n, k], ref.constituent_dict[f'c{k}'].state[steps][:n]) <= TOL_CONC


@pytest.mark.mid_mesh_default
@pytest.mark.parametrize('parts', [2, 6, 12])
def test_several_parts_on_a_river_band_with_dry_cells_at_the_ohio_time_step(gpu_lib, parts, monkeypatch):
    """one case per part count on the reference's everyday regime -- 75 m cells, dt = 3600 s, 0.5 % dry cells,
    breathing volumes -- through k_small_jacobi<RPT, true>: element-wise against the oracle after every step, no flag, no warning."""
    import warnings
    import clearwater_riverine_amd as cw
    K, steps = 2, 6
    nx, ny = {2: (110, 50), 6: (200, 50), 12: (300, 60)}[parts]
    mesh = cw.synthetic.make_mesh(nx, ny, steps, seed=20100529, n_merge=nx * ny // 40, dx=75.0, dy=75.0, depth=3.0, dt=3600.0, velocity=0.3,
                                  breathing=0.1, diffusion_coefficient=0.1, period_steps=24, n_dry=nx * ny // 200)
    oracle.derive_coefficients(mesh)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=86400.0)
    n = mesh['nreal'] + 1
    monkeypatch.setenv('CWR_SMALL_PARTS', str(parts))
    ref = oracle_run(mesh, inputs3, steps)
    eng = make_engine(mesh, inputs3)
    eng.set_state(inputs3[0, :n, :])
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        for t in range(steps):
            r = eng.step(t, tol=1e-12)
            assert r.sweep_kernel == 7 and r.flags == 0 and r.iterations == 0
            got = eng.get_state()
            for k in range(K):
                assert rel_err(got[:n, k], ref.constituent_dict[f'c{k}'].state[t + 1][:n]) <= TOL_CONC
    eng.close()


@pytest.mark.parametrize('nx,ny,K', [(1, 1, 1), (2, 1, 2), (1, 5, 3), (2, 2, 1), (3, 3, 64), (64, 1, 2), (1, 70, 1), (33, 31, 5)])
def test_tiny_and_degenerate_meshes_match_the_oracle(gpu_lib, nx, ny, K):
    """One cell, one row, one column, 64 constituents on nine cells: the one-launch solver's tables (a wave with a single live row,
    empty row slots, rows without a real neighbour) against the oracle."""
    import clearwater_riverine_amd as cw
    mesh, inputs3 = synthetic_case(K, nx=nx, ny=ny, n_steps=4, seed=3, n_merge=0, dt=30.0, diffusion_coefficient=0.2)
    names = [f'c{k}' for k in range(K)]
    ref = oracle_run(mesh, inputs3, 4)
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)})
    for _ in range(4):
        model.update()
        assert model.last_step.max_rel_residual <= 1e-12 and model.last_step.sweep_kernel == 7
    for nm in names[:3] + names[-1:]:
        assert rel_err(model.mesh[nm], ref.constituent_dict[nm].state) <= TOL_CONC
