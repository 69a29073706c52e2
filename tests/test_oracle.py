"""CPU tests of the oracle: pinned against the reference's own fixture facts and notebook-printed
known answers, against the committed expected outputs, and against the
identities the reference's algebra obeys."""
import os

import numpy as np
import pytest

import cwr_oracle as oracle
from util import GOLDEN, load_plan, multi_inputs


def test_fixture_topology_facts():
    """plan01 80 cells / 50 real / 115 faces, plan02-03 8 cells / 2 real / 7 faces;
    HDF dtypes float32 flows/volumes, int32 indices; ghosts appear only as face2."""
    z1 = np.load(os.path.join(GOLDEN, 'plan01_inputs.npz'))
    assert len(z1['face_x']) == 80 and z1['edges_face1'].max() == 49 and len(z1['edges_face1']) == 115
    assert int(z1['n_time_stamps_in_hdf']) == 10801
    counts = np.bincount(z1['edges_face2'][z1['edges_face2'] > 49], minlength=80)[50:]
    assert np.all(counts == 1)                               # every ghost cell is used by exactly one face
    assert not np.all(np.diff(z1['edges_face1']) >= 0)       # face1 is not sorted
    for plan, nstamps in (('plan02', 25), ('plan03', 7201)):
        z = np.load(os.path.join(GOLDEN, f'{plan}_inputs.npz'))
        assert len(z['face_x']) == 8 and z['edges_face1'].max() == 1 and len(z['edges_face1']) == 7
        assert int(z['n_time_stamps_in_hdf']) == nstamps
        assert z['face_flow'].dtype == np.float32 and z['volume'].dtype == np.float32
        assert z['edges_face1'].dtype == np.int32


def test_time_stamp_counts_of_reference_test():
    """reference tests/test_riverine.py:78-86: 25 stamps; (5, 8) -> 4; 12:00..12:10 -> 3."""
    z = np.load(os.path.join(GOLDEN, 'plan02_inputs.npz'))
    ts = oracle.parse_ras_stamps(z['time_stamps'])
    assert len(ts) == 25
    assert len(ts[5:8 + 1]) == 4                             # io/hdf.py:160-164: inclusive int range
    assert np.count_nonzero((ts >= 0.0) & (ts <= 600.0)) == 3
    assert np.allclose(np.diff(ts), 300.0)


def test_boundary_ghost_cells_of_reference_test():
    """reference tests/test_riverine.py:101-106: IC cell 0 = 100; BC lands in ghost cells 4 and 6."""
    mesh, inp, z = load_plan('plan02', 0.01)
    assert list(z['bc_line_names']) == ['US_Flow', 'DS_Stage']
    assert list(z['bc_face_index']) == [3, 5]
    ghosts = mesh['edges_face2'][z['bc_face_index']]
    assert list(ghosts) == [4, 6] and list(mesh['edges_face1'][z['bc_face_index']]) == [0, 1]
    assert inp[0, 0] == 100.0 and np.all(inp[:, 4] == 100.0) and np.all(inp[:, 6] == 100.0)
    model = oracle.OracleModel(mesh, {'c': inp})
    model.update()                                           # :109-127 after one update
    st = model.constituent_dict['c'].state
    assert model.time_step == 1 and st[1, 0] != 0 and st[1, 4] == 100.0 and st[1, 6] == 100.0


def test_notebook_mass_known_answer():
    """examples/dev_sandbox/test_functions_for_pytest.ipynb cell[1]: 5000.553131 / 5001.221848."""
    z = np.load(os.path.join(GOLDEN, 'plan02_inputs.npz'))
    assert (z['volume'][0, :2].astype(np.float64) * 100).sum() == pytest.approx(5000.553131, abs=5e-6)
    assert (z['volume_last'][:2].astype(np.float64) * 100).sum() == pytest.approx(5001.221848, abs=5e-6)


def test_notebook_diffusion_sums_known_answer():
    """examples/dev_sandbox/49_sum_coef_diffusion.ipynb cell[35-36]: plan03, D = 0.001, t = 1."""
    mesh, _, _ = load_plan('plan03', 0.001)
    d1 = mesh['coeff_to_diffusion'][1]
    sums = np.zeros(8)
    np.add.at(sums, mesh['edges_face1'], d1)
    np.add.at(sums, mesh['edges_face2'], d1)
    want = [0.00300032, 0.00300033, 0, 0, 0.00200021, 0, 0.00200022, 0]
    assert np.allclose(sums, want, atol=5e-9)
    assert np.allclose(mesh['face_to_face_dist'][:2], [5.0, 2.5])


@pytest.mark.parametrize('plan,D', [('plan01', 0.01), ('plan02', 0.01), ('plan03', 0.001)])
def test_committed_expected_outputs(plan, D):
    """The oracle reproduces the committed expected outputs bit for bit (regression pin)."""
    exp = np.load(os.path.join(GOLDEN, f'{plan}_expected.npz'))
    steps = int(exp['steps'])
    mesh, inp, _ = load_plan(plan, D)
    assert np.array_equal(mesh['advection_coeff'][:steps + 1], exp['advection_coeff'])
    assert np.array_equal(mesh['coeff_to_diffusion'][:steps + 1], exp['coeff_to_diffusion'])
    model = oracle.OracleModel(mesh, {'c': inp})
    for _ in range(steps):
        model.update()
    con = model.constituent_dict['c']
    assert np.allclose(con.state[:steps + 1], exp['state'], rtol=1e-12, atol=0, equal_nan=True)
    assert np.allclose(con.total_mass_flux[:steps], exp['total_mass_flux'], rtol=1e-10, atol=1e-12, equal_nan=True)


def test_smoke_values_of_the_survey_session():
    """SURVEY 8c restatement self-consistency values (plan01 to all printed digits; plan02 to 2e-8: the
    survey's throw-away restatement multiplied area * D in float64, the reference does it in float32)."""
    mesh, inp, _ = load_plan('plan01', 0.01)
    m = oracle.OracleModel(mesh, {'c': inp})
    m.update()
    c = m.constituent_dict['c'].state[1, :50]
    assert c.min() == pytest.approx(99.99971393017036, rel=1e-13) and c.max() == pytest.approx(102.01319390332685, rel=1e-13)
    assert m.last_A.nnz == 220
    mesh, inp, _ = load_plan('plan02', 0.01)
    m = oracle.OracleModel(mesh, {'c': inp})
    m.update()
    c = m.constituent_dict['c'].state[1, :2]
    assert c.min() == pytest.approx(124.01039581449467, rel=5e-8) and c.max() == pytest.approx(174.0077918243957, rel=5e-8)
    assert m.last_A.nnz == 4


@pytest.mark.parametrize('plan,D', [('plan01', 0.01), ('plan02', 0.01), ('plan03', 0.001), ('plan01', 0.0)])
def test_coo_assembly_equals_percell_form(plan, D):
    """linalg.py:34-156 entry by entry == the per-cell face-flux form the HIP kernel implements."""
    mesh, _, _ = load_plan(plan, D)
    n = mesh['nreal'] + 1
    lhs = oracle.LHS(mesh)
    rng = np.random.default_rng(0)
    for t in (0, 1, 5, 20):
        lhs.update_values(mesh, t)
        A = lhs.csr()
        for K in (1, 3):
            x = rng.standard_normal((n, K))
            ref = A @ x
            assert np.max(np.abs(oracle.apply_percell(mesh, t, x) - ref)) <= 1e-13 * np.max(np.abs(ref))
        # the COO arrays keep the reference's layout: float rows/cols, over-allocated zero tail
        assert lhs.rows.dtype == np.float64 and lhs.coef.dtype == np.float64
        assert len(lhs.coef) == 2 * lhs.internal_edge_count + 2 * n + 2 * np.count_nonzero(mesh['advection_coeff'][t] > 0) + \
            2 * np.count_nonzero((mesh['advection_coeff'][t] < 0) & np.isin(np.arange(len(lhs.rows) * 0 + len(mesh['edges_face1'])), lhs.internal_edges)) + \
            np.count_nonzero(mesh['volume'][t + 1][:n] == 0) + len(lhs.real_edges_face1) + len(lhs.real_edges_face2)


def test_rhs_literal_equals_percell_form_incl_last_write_wins():
    import clearwater_riverine_amd as cw
    mesh = cw.synthetic.make_mesh(12, 7, 6, seed=5, n_merge=6)
    oracle.derive_coefficients(mesh)
    inputs3 = cw.synthetic.boundary_input_array(mesh, 2)
    n = mesh['nreal'] + 1
    # corner cells own two ghost faces: the literal restatement's assignment keeps the higher face id
    f1, f2 = mesh['edges_face1'], mesh['edges_face2']
    ghost_faces_per_cell = np.bincount(f1[f2 > mesh['nreal']], minlength=n)
    assert ghost_faces_per_cell.max() >= 2
    rng = np.random.default_rng(3)
    for t in (1, 3):
        x = 1.0 + rng.random(n)
        for k in range(2):
            r = oracle.RHS(mesh, inputs3[:, :, k].copy())
            r.update_values(x, mesh, t)
            want = oracle.rhs_percell(mesh, t, x, inputs3[t + 1, :, k])
            assert np.max(np.abs(r.vals - want)) <= 1e-13 * np.max(np.abs(want))


def test_zero_coefficient_on_active_ghost_face_is_a_value_error():
    """linalg.py:349-351: the `!= 0` filter makes the assignment a shape mismatch."""
    import clearwater_riverine_amd as cw
    mesh = cw.synthetic.make_mesh(8, 5, 3, seed=1)
    oracle.derive_coefficients(mesh)
    inp = cw.synthetic.boundary_input_array(mesh, 1)[:, :, 0]
    f2 = mesh['edges_face2']
    inlet_faces = np.nonzero(np.isin(f2, mesh['inlet_ghost_cells']))[0]
    assert len(inlet_faces) >= 3
    mesh['advection_coeff'][2, inlet_faces[:2]] = 0.0          # velocity still < 0 there
    r = oracle.RHS(mesh, inp)
    with pytest.raises(ValueError):
        r.update_values(np.ones(mesh['nreal'] + 1), mesh, 1)


def test_mass_identity_on_steady_flow():
    """With a steady, discretely divergence-free field the scheme is conservative:
    sum(V c)[t+1] - sum(V c)[t] = sum over boundary faces of (diffusion_mass_flux - advection_mass_flux).
    (transport.py:419-427 signs: the advective flux is positive from face1 to face2, the diffusive one
    d*(c_N - c_P) is positive INTO face1, so `total = advection + diffusion` mixes the two conventions;
    the budget needs their difference.)"""
    import clearwater_riverine_amd as cw
    mesh = cw.synthetic.make_mesh(20, 10, 40, seed=1, steady=True, shuffle_window=8)
    oracle.derive_coefficients(mesh)
    inp = cw.synthetic.boundary_input_array(mesh, 1)[:, :, 0]
    inp[:, mesh['outlet_ghost_cells']] = 5.0                   # every open boundary has a value
    model = oracle.OracleModel(mesh, {'c': inp})
    n = mesh['nreal'] + 1
    ghost = mesh['edges_face2'] > mesh['nreal']
    for s in range(40):
        model.update()
        con = model.constituent_dict['c']
        V0 = mesh['volume'][s, :n].astype(np.float64)
        V1 = mesh['volume'][s + 1, :n].astype(np.float64)
        lhs = (V1 * con.state[s + 1, :n]).sum() - (V0 * con.state[s, :n]).sum()
        rhs = np.nansum(con.diffusion_mass_flux[s][ghost]) - np.nansum(con.advection_mass_flux[s][ghost])
        assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), abs(rhs), (V1 * con.state[s + 1, :n]).sum() * 1e-3)


def test_override_and_initial_condition_quirk():
    """transport.py:233-236 override, and linalg.py:199-200: at t = 0 the non-zero IC entries overwrite it."""
    mesh, inp, _ = load_plan('plan02', 0.01)
    a = oracle.OracleModel(dict(mesh), {'c': inp.copy()})
    a.update({'c': np.array([7.0, 9.0])})
    b = oracle.OracleModel(dict(mesh), {'c': inp.copy()})
    b.update()
    assert np.array_equal(a.constituent_dict['c'].state[1], b.constituent_dict['c'].state[1], equal_nan=True)
    a.update({'c': np.array([7.0, 9.0])})
    b.update()
    assert not np.allclose(a.constituent_dict['c'].state[2, :2], b.constituent_dict['c'].state[2, :2])
    assert np.array_equal(a.constituent_dict['c'].state[1, :2], [7.0, 9.0])
