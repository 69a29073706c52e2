"""GPU tests (through the C ABI): committed golden outputs, error behaviour mirroring the reference,
size-independent properties at large sizes, run-to-run reproducibility."""
import os

import numpy as np
import pytest

import cwr_oracle as oracle
from util import GOLDEN, flux_err, load_plan, rel_err
from test_gpu_parity import make_engine, synthetic_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('plan', ['plan01', 'plan02', 'plan03'])
def test_against_committed_golden_outputs(gpu_lib, plan):
    """HIP path vs tests/golden/*_expected.npz (oracle outputs on the reference's HDF fixtures)."""
    import clearwater_riverine_amd as cw
    exp = np.load(os.path.join(GOLDEN, f'{plan}_expected.npz'))
    D, steps = float(exp['diffusion_coefficient']), int(exp['steps'])
    mesh, inp, _ = load_plan(plan, D)
    model = cw.ClearwaterRiverine(mesh=dict(mesh), diffusion_coefficient_input=D, input_arrays={'c': inp.copy()})
    for _ in range(steps):
        model.update()
    assert rel_err(model.mesh['c'][:steps + 1], exp['state']) <= 1e-9
    assert flux_err(model.constituent_dict['c'].advection_mass_flux[:steps], exp['advection_mass_flux']) <= 1e-8
    assert flux_err(model.constituent_dict['c'].diffusion_mass_flux[:steps], exp['diffusion_mass_flux']) <= 1e-8
    adv, dif = model.coefficients(3)
    assert np.array_equal(adv, exp['advection_coeff'][3]) and np.array_equal(dif, exp['coeff_to_diffusion'][3])


def test_zero_coefficient_on_active_ghost_face_raises_value_error(gpu_lib):
    """Mirrors the reference's shape-mismatch ValueError (linalg.py:349-351) for rhs() and step()."""
    import clearwater_riverine_amd as cw
    mesh, inputs3 = synthetic_case(1, nx=8, ny=5, n_steps=3, seed=1)
    f2 = mesh['edges_face2']
    inlet_faces = np.nonzero(np.isin(f2, mesh['inlet_ghost_cells']))[0]
    flow = mesh['face_flow'].copy()
    flow[2, inlet_faces[:2]] = 0.0                       # advection_coeff becomes 0 while velocity stays < 0
    n = mesh['nreal'] + 1
    eng = cw.TransportEngine(mesh['edges_face1'], f2, len(mesh['face_x']), 1)
    eng.load_flow_field(flow, mesh['edge_velocity'], mesh['volume'], mesh['dt'], mesh['face_to_face_dist'], 0.1)
    eng.load_boundary(inputs3[:, n:, :])
    eng.set_state(np.ones((n, 1)))
    with pytest.raises(ValueError, match='ghost face'):
        eng.rhs(1, np.ones((n, 1)))
    with pytest.raises(ValueError, match='ghost face'):
        eng.step(1)
    eng.step(0)                                           # level 1 boundary terms are fine


def test_call_order_and_range_errors(gpu_lib):
    import clearwater_riverine_amd as cw
    mesh, inputs3 = synthetic_case(2, nx=6, ny=4, n_steps=2, seed=0)
    n = mesh['nreal'] + 1
    eng = cw.TransportEngine(mesh['edges_face1'], mesh['edges_face2'], len(mesh['face_x']), 2)
    with pytest.raises(IndexError, match='no flow field'):
        eng.step(0)
    eng.load_flow_field(mesh['face_flow'], mesh['edge_velocity'], mesh['volume'], mesh['dt'], mesh['face_to_face_dist'], 0.1)
    with pytest.raises(IndexError, match='boundary'):
        eng.step(0)
    eng.load_boundary(inputs3[:, n:, :])
    with pytest.raises(IndexError, match='out of range'):
        eng.step(2)                                       # needs level 3
    with pytest.raises(ValueError):
        eng.set_state(np.ones((n + 1, 2)))                # wrapper shape check
    with pytest.raises(ValueError):
        eng.step(0, tol=0.0)


def test_not_converged_is_reported(gpu_lib):
    import clearwater_riverine_amd as cw
    mesh, inputs3 = synthetic_case(1, nx=40, ny=20, n_steps=2, seed=2)
    eng = make_engine(mesh, inputs3)
    eng.set_state(inputs3[0, :mesh['nreal'] + 1, :])
    for solver in ('auto', 'jacobi', 'bicgstab'):
        eng.set_state(inputs3[0, :mesh['nreal'] + 1, :])
        with pytest.raises(cw.SolverNotConverged):
            eng.step(0, tol=1e-14, max_iter=3, solver=solver)


def test_nan_state_is_reported_not_looped(gpu_lib):
    mesh, inputs3 = synthetic_case(1, nx=10, ny=6, n_steps=2, seed=2)
    eng = make_engine(mesh, inputs3)
    for solver in ('auto', 'bicgstab'):
        x = inputs3[0, :mesh['nreal'] + 1, :].copy()
        x[3] = np.nan
        eng.set_state(x)
        with pytest.raises(FloatingPointError):
            eng.step(0, solver=solver)


@pytest.mark.parametrize('limit', ['100', None])
def test_stiff_step_and_the_handover_to_bicgstab(gpu_lib, monkeypatch, limit):
    """Large CFL (dt = 3600 s on 10 m cells, beyond the Ohio River regime).  By default the block-asynchronous passes
    finish it on their own (measured: 132 sweep equivalents, 15 x faster than BiCGSTAB); with a sweep budget
    (CWR_JACOBI_LIMIT) the measured contraction predicts too many sweeps and the step is handed to BiCGSTAB from the
    current iterate.  Either way the result matches the direct solve."""
    import clearwater_riverine_amd as cw
    if limit:
        monkeypatch.setenv('CWR_JACOBI_LIMIT', limit)
    mesh = cw.synthetic.make_mesh(160, 40, 3, seed=3, dt=3600.0, breathing=0.0, n_merge=10)   # > 4096 cells: multi-launch path
    oracle.derive_coefficients(mesh)
    inputs3 = cw.synthetic.boundary_input_array(mesh, 2, inlet_period_s=86400.0)
    n = mesh['nreal'] + 1
    eng = make_engine(mesh, inputs3)
    eng.set_state(inputs3[0, :n, :])
    res = eng.step(0)
    if limit:
        assert res.solver == 2 and res.iterations > 0
    else:
        assert res.solver == 0 and res.iterations == 0 and res.sweeps > 50
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(2)})
    ref.update()
    got = eng.get_state()
    for k in range(2):
        assert rel_err(got[:, k], ref.constituent_dict[f'c{k}'].state[1]) <= 1e-9


def test_load_coefficients_route_equals_device_derivation(gpu_lib):
    """cwr_load_coefficients (the reference's already-derived Dataset variables) == cwr_load_flow_field."""
    import clearwater_riverine_amd as cw
    mesh, inputs3 = synthetic_case(3, nx=16, ny=9, n_steps=4, seed=8, n_merge=10)
    n = mesh['nreal'] + 1
    a = make_engine(mesh, inputs3)
    b = cw.TransportEngine(mesh['edges_face1'], mesh['edges_face2'], len(mesh['face_x']), 3)
    b.load_coefficients(mesh['advection_coeff'], mesh['coeff_to_diffusion'], mesh['edge_velocity'], mesh['volume'],
                        mesh['dt'], mesh['diffusion_coefficient'])
    b.load_boundary(inputs3[:, n:, :])
    for e in (a, b):
        e.set_state(inputs3[0, :n, :])
        for t in range(3):
            e.step(t)
    assert np.array_equal(a.get_state(), b.get_state(), equal_nan=True)


def test_run_to_run_bitwise_reproducible(gpu_lib):
    """Inner products are reduced in a fixed order (no float atomics): two runs agree bit for bit."""
    mesh, inputs3 = synthetic_case(4, nx=120, ny=50, n_steps=3, seed=6, n_merge=200)
    n = mesh['nreal'] + 1
    outs = []
    for _ in range(2):
        eng = make_engine(mesh, inputs3)
        eng.set_state(inputs3[0, :n, :])
        for t in range(3):
            eng.step(t)
        outs.append(eng.get_state())
        eng.close()
    assert np.array_equal(outs[0], outs[1], equal_nan=True)


@pytest.mark.parametrize('K', [1, 16])
def test_full_size_properties_1m_cells(gpu_lib, K):
    """BASELINE.json's 1 M-cell mesh: size-independent properties instead of a direct solve.
    (i) linearity of the operator, (ii) the solve's true residual, (iii) constant-state preservation rows:
    A.1 = V[t+1]/dt + net outflow, (iv) scaling invariance across constituents."""
    import clearwater_riverine_amd as cw
    mesh = cw.synthetic.make_mesh(1000, 1000, 2, seed=4, dt=40.0, diffusion_coefficient=0.5)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K)
    n = mesh['nreal'] + 1
    from clearwater_riverine_amd.model import face_to_face_distance, change_in_time
    mesh['face_to_face_dist'] = face_to_face_distance(mesh)
    mesh['dt'] = change_in_time(mesh['time_seconds'])
    eng = make_engine(mesh, inputs3)
    rng = np.random.default_rng(0)
    x1, x2 = rng.standard_normal((n, K)), rng.standard_normal((n, K))
    y1, y2, y12 = eng.apply(0, x1), eng.apply(0, x2), eng.apply(0, 2.0 * x1 - 3.0 * x2)
    assert np.max(np.abs(y12 - (2.0 * y1 - 3.0 * y2))) <= 1e-12 * np.max(np.abs(y1))
    # row sums against an independent numpy evaluation from the raw fields
    adv, dif = eng.get_coefficients(0)
    f1, f2 = mesh['edges_face1'].astype(np.int64), mesh['edges_face2'].astype(np.int64)
    a = adv.astype(np.float64)
    internal = f2 < n
    rows = mesh['volume'][1, :n].astype(np.float64) / mesh['dt'][0]
    rows += np.bincount(f1, weights=np.where(internal, a, np.maximum(a, 0.0) + dif), minlength=n)
    rows -= np.bincount(f2[internal], weights=a[internal], minlength=n)
    ones = eng.apply(0, np.ones((n, K)))
    assert np.max(np.abs(ones[:, 0] - rows)) <= 1e-11 * np.max(np.abs(rows))
    # one implicit step: residual of the returned state, through the exported operator and right-hand side
    x0 = inputs3[0, :n, :]
    eng.set_state(x0)
    b = eng.rhs(0, x0)
    res = eng.step(0, tol=1e-12)
    xs = eng.get_state()[:n]
    r = b - eng.apply(0, xs)
    assert np.max(np.linalg.norm(r, axis=0) / np.linalg.norm(b, axis=0)) <= 1e-10
    assert res.sweeps + res.iterations < 200
    # the other solver reaches the same state
    eng.set_state(x0)
    eng.step(0, tol=1e-12, solver='bicgstab')
    assert np.max(np.abs(eng.get_state()[:n] - xs)) <= 1e-9 * np.max(np.abs(xs))
    if K > 1:                                              # constituent k is (k+1) x constituent 0
        assert np.max(np.abs(xs[:, K - 1] - K * xs[:, 0])) <= 1e-9 * np.max(np.abs(xs[:, K - 1]))


def test_single_rank_rccl_communicator(gpu_lib, monkeypatch):
    """The RCCL plumbing that can run on one GPU: dlopen of librccl, unique id, ncclCommInitRank with one rank,
    and (CWR_FORCE_COLLECTIVES=1) the all-reduce call sites inside both solver loops.  Results must equal the
    communicator-free engine bit for bit (an all-reduce over one rank is the identity)."""
    import clearwater_riverine_amd as cw
    monkeypatch.setenv('CWR_FORCE_COLLECTIVES', '1')
    monkeypatch.setenv('CWR_NO_SMALL', '1')          # both engines on the multi-launch path (a communicator rules out the small one)
    monkeypatch.setenv('CWR_TWO_CLOSING', '1')       # ... and the same batch shape (engines with a communicator count sweeps in pairs)
    mesh, inputs3 = synthetic_case(3, nx=30, ny=14, n_steps=3, seed=12, n_merge=15)
    n = mesh['nreal'] + 1
    outs = []
    for with_comm in (False, True):
        eng = make_engine(mesh, inputs3)
        if with_comm:
            uid = cw.TransportEngine.comm_unique_id()
            assert len(uid) == 128
            eng.attach_comm(0, 1, uid, [], [0], [], [0], [])
            # the real librccl's point-to-point entry points (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd) on the
            # communication stream, this rank to itself: the signatures and the event plumbing of the overlapped exchange
            assert eng.comm_selftest(4096) == 0
        eng.set_state(inputs3[0, :n, :])
        eng.step(0, solver='jacobi')
        eng.step(1, solver='bicgstab')
        outs.append(eng.get_state())
        eng.close()
    assert np.array_equal(outs[0], outs[1], equal_nan=True)


def test_device_reaction_equals_host_callback_and_oracle(gpu_lib):
    """the in-HBM reaction step (cwr_react_linear) == the host callback through
    update_concentration (transport.py:233-236) == the oracle driven with the same override."""
    import clearwater_riverine_amd as cw
    K = 4
    mesh, inputs3 = synthetic_case(K, nx=24, ny=10, n_steps=8, seed=13, n_merge=12)
    names = [f'c{k}' for k in range(K)]
    n = mesh['nreal'] + 1
    lam = np.array([0.0, 1e-3, 5e-4, 2e-3])
    M = np.diag(np.exp(-lam * 10.0))
    M[1, 0] = 0.01                                       # pairwise exchange: c1 gains from c0, c0 loses the same amount
    M[0, 0] -= 0.01
    arrays = {nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)}
    dev = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={k: v.copy() for k, v in arrays.items()})
    host = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={k: v.copy() for k, v in arrays.items()})
    ref = oracle.OracleModel(mesh, {k: v.copy() for k, v in arrays.items()})
    for s in range(8):
        if s == 0:
            dev.update(); host.update(); ref.update()
            continue
        dev.update(reaction_matrix=M)
        c = np.stack([host.mesh[nm][s][:n] for nm in names], axis=1)
        upd = {nm: (c @ M.T)[:, k] for k, nm in enumerate(names)}
        host.update(upd)
        c = np.stack([ref.constituent_dict[nm].state[s][:n] for nm in names], axis=1)
        ref.update({nm: (c @ M.T)[:, k] for k, nm in enumerate(names)})
    for nm in names:
        assert rel_err(dev.mesh[nm], host.mesh[nm]) <= 1e-12
        assert rel_err(dev.mesh[nm], ref.constituent_dict[nm].state) <= 1e-9
    ptr, stream = dev.engine.state_device_ptr()
    assert ptr and stream


@pytest.mark.parametrize('local_reps', ['1', 'default'])
def test_internal_hilbert_renumbering_is_transparent(gpu_lib, monkeypatch, local_reps):
    """ordering.py: with exact Jacobi passes (CWR_LOCAL_REPS=1) results in the reference's numbering are bitwise
    identical with and without the internal space-filling-curve renumbering (row sums visit faces in ascending
    face id either way); the default block-asynchronous passes depend on the tiling and agree to solver tolerance."""
    from clearwater_riverine_amd.distributed import PartitionedTransport
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    if local_reps == '1':
        monkeypatch.setenv('CWR_LOCAL_REPS', '1')
    mesh, inputs3 = synthetic_case(8, nx=60, ny=33, n_steps=3, seed=17, n_merge=80)
    outs = []
    for ren in (None, 'hilbert'):
        pt = PartitionedTransport(mesh, inputs3, 0, 1, renumber=ren)
        for t in range(3):
            pt.step(t, solver='jacobi')
        outs.append(pt.gather_state())
        assert np.array_equal(np.sort(pt.owned_reference_ids()), np.arange(mesh['nreal'] + 1))
    if local_reps == '1':
        assert np.array_equal(outs[0], outs[1])
    assert rel_err(outs[0], outs[1]) <= 1e-10
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(8)})
    for _ in range(3):
        ref.update()
    n = mesh['nreal'] + 1
    want = np.stack([ref.constituent_dict[f'c{k}'].state[3, :n] for k in range(8)], axis=1)
    assert rel_err(outs[1], want) <= 1e-9


def test_engine_cell_order_is_transparent(gpu_lib, monkeypatch):
    """TransportEngine(cell_order=...): apply / rhs / step / get_state in the reference's numbering are unchanged
    (bitwise, with exact Jacobi passes)."""
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.ordering import hilbert_order
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_LOCAL_REPS', '1')
    K = 8
    mesh, inputs3 = synthetic_case(K, nx=50, ny=31, n_steps=3, seed=19, n_merge=40, n_dry=2)
    n = mesh['nreal'] + 1
    ncell = len(mesh['face_x'])
    order = hilbert_order(mesh['face_x'], mesh['face_y'], n)
    engs = []
    for o in (None, order):
        e = cw.TransportEngine(mesh['edges_face1'], mesh['edges_face2'], ncell, K, cell_order=o)
        e.load_flow_field(mesh['face_flow'], mesh['edge_velocity'], mesh['volume'], mesh['dt'], mesh['face_to_face_dist'],
                          mesh['diffusion_coefficient'])
        e.load_boundary(inputs3[:, n:, :])
        engs.append(e)
    x = np.random.default_rng(3).standard_normal((n, K))
    assert np.array_equal(engs[0].apply(1, x), engs[1].apply(1, x))
    assert np.array_equal(engs[0].rhs(1, x), engs[1].rhs(1, x))
    for e in engs:
        e.set_state(inputs3[0, :n, :])
        for t in range(3):
            e.step(t, solver='jacobi')
    assert np.array_equal(engs[0].get_state(), engs[1].get_state(), equal_nan=True)
    a, b = engs[0].get_mass_flux(), engs[1].get_mass_flux()
    assert all(np.array_equal(p, q, equal_nan=True) for p, q in zip(a, b))
    with pytest.raises(ValueError):
        cw.TransportEngine(mesh['edges_face1'], mesh['edges_face2'], ncell, K, cell_order=order[:-1])


@pytest.mark.parametrize('K', [1, 16])
def test_block_asynchronous_passes_converge_faster_to_the_same_solution(gpu_lib, monkeypatch, K):
    """k_sq_tiled with reps > 1 (tile-local re-application of J^2): fewer sweeps than exact Jacobi passes, same
    solution as the oracle's spsolve (<= 1e-9, bar 1e-6), exact residual check unchanged, run-to-run bitwise."""
    from clearwater_riverine_amd.distributed import PartitionedTransport
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    mesh, inputs3 = synthetic_case(K, nx=120, ny=60, n_steps=4, seed=23, n_merge=100, dt=40.0, diffusion_coefficient=0.5)
    n = mesh['nreal'] + 1
    sweeps, outs = {}, {}
    for reps in ('1', '2', '2', '3'):
        monkeypatch.setenv('CWR_LOCAL_REPS', reps)
        pt = PartitionedTransport(mesh, inputs3, 0, 1, renumber='hilbert')
        res = [pt.step(t, tol=1e-12, solver='jacobi') for t in range(4)]
        assert all(r.max_rel_residual <= 1e-12 for r in res)
        assert res[-1].sweep_kernel == 6                      # the tiled pass is what ran
        if reps in outs:
            assert np.array_equal(outs[reps], pt.gather_state())   # deterministic
        sweeps[reps] = res[-1].sweeps
        outs[reps] = pt.gather_state()
    assert sweeps['2'] < sweeps['1'] and sweeps['3'] <= sweeps['2']
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for _ in range(4):
        ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[4, :n] for k in range(K)], axis=1)
    for reps in outs:
        assert rel_err(outs[reps], want) <= 1e-9


@pytest.mark.parametrize('K', [1, 16])
def test_persistent_grid_size_of_the_tiled_pass_does_not_change_the_result(gpu_lib, monkeypatch, K):
    """The tiled pass is a persistent launch whose blocks walk a static share of the tiles (csrc launch_sq_tiled): with
    8 blocks, with one resident block per CU and with every slot taken the state is bitwise the same (the pass reads
    one buffer and writes the other, so which block computes a tile cannot matter)."""
    from clearwater_riverine_amd.distributed import PartitionedTransport
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    # (the ping-pong passes: with 8 blocks for 113 tiles the engine would otherwise chain the tiles along the flow and relax
    # in place, where which block computes a tile DOES matter in the last bits -- tests/test_gpu_chains.py)
    monkeypatch.setenv('CWR_NO_CHAINS', '1')
    mesh, inputs3 = synthetic_case(K, nx=120, ny=60, n_steps=3, seed=29, n_merge=150, dt=40.0, diffusion_coefficient=0.5)
    outs = []
    for knob, val in ((None, None), ('CWR_TCL_GRID', '8'), ('CWR_TCL_BLOCKS_PER_CU', '1')):
        if knob:
            monkeypatch.setenv(knob, val)
        pt = PartitionedTransport(mesh, inputs3, 0, 1, renumber='hilbert')
        res = [pt.step(t, tol=1e-12, solver='jacobi') for t in range(3)]
        assert res[-1].sweep_kernel == 6 and all(r.max_rel_residual <= 1e-12 for r in res)
        outs.append(pt.gather_state())
        if knob:
            monkeypatch.delenv(knob)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


@pytest.mark.parametrize('K', [1, 16])
def test_batch_shapes_agree_and_the_one_sweep_shape_never_ends_outside_the_state_vector(gpu_lib, monkeypatch, K):
    """One GPU: a batch is N J^2 passes + one closing sweep, its parity fixed by starting from the kept copy of x_t
    (csrc solve_jacobi); CWR_TWO_CLOSING=1 is round 1's shape (even passes + two sweeps).  Both must give the oracle's
    answer; with exact passes (CWR_LOCAL_REPS=1) every iterate is a plain Jacobi iterate, so the two runs differ only by
    how many sweeps they took and agree to the solver tolerance; odd and even sweep counts both occur."""
    from clearwater_riverine_amd.distributed import PartitionedTransport
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_LOCAL_REPS', '1')
    mesh, inputs3 = synthetic_case(K, nx=120, ny=60, n_steps=6, seed=37, n_merge=150, dt=40.0, diffusion_coefficient=0.5)
    n = mesh['nreal'] + 1
    runs = {}
    for shape in ('one', 'two'):
        if shape == 'two':
            monkeypatch.setenv('CWR_TWO_CLOSING', '1')
        pt = PartitionedTransport(mesh, inputs3, 0, 1, renumber='hilbert')
        res = [pt.step(t, tol=1e-12, solver='jacobi') for t in range(6)]
        assert all(r.max_rel_residual <= 1e-12 and r.sweep_kernel == 6 for r in res)
        runs[shape] = (pt.gather_state(), [r.sweeps for r in res])
    assert all(s % 2 == 0 for s in runs['two'][1])                      # even passes + two sweeps
    assert any(s % 2 == 1 for s in runs['one'][1])                      # N passes + one sweep
    assert sum(runs['one'][1]) <= sum(runs['two'][1])
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for _ in range(6):
        ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[6, :n] for k in range(K)], axis=1)
    for shape in runs:
        assert rel_err(runs[shape][0], want) <= 1e-9


@pytest.mark.parametrize('K', [4, 16])
def test_the_passes_batch_sizes_do_not_run_away_at_cfl_18(gpu_lib, K):
    """An 18 k-cell river band at the reference's own time step (75 m cells, dt = 3600 s, CFL ~ 18) through the tiled passes (the
    conftest keeps mid-size meshes on them), 40 steps.  Round 5 found the batch-size prediction running away here: a check that
    missed only the element-wise rule was extrapolated geometrically (hundreds of sweeps too many), and two checks at the rounding
    floor measured a contraction of 0.9999 that sized the next batch at the sweep limit -- 163, 832, 593, 353 ... 2002 sweeps a step
    where ~170 do.  With the rate clamped by ||J||_inf and the follow-up batches bounded no step takes more than 1.6 x the median."""
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.distributed import PartitionedTransport
    T = 44
    mesh = cw.synthetic.make_mesh(300, 60, T, seed=20100529, n_merge=0, dx=75.0, dy=75.0, depth=3.0, dt=3600.0, velocity=0.3, breathing=0.0,
                                  diffusion_coefficient=0.1, period_steps=24)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=86400.0)
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    sweeps = []
    for t in range(40):
        r = pt.engine.step(t, tol=1e-12)
        assert r.sweep_kernel == 6 and r.flags == 0 and r.iterations == 0 and r.max_rel_residual <= 1e-12
        sweeps.append(r.sweeps)
    pt.engine.close()
    med = float(np.median(sweeps))
    assert max(sweeps[1:]) <= 1.6 * med, sweeps
