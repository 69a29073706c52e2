"""CPU tests of the host-side logic around the C ABI (no GPU compute)."""
import os

import numpy as np
import pandas as pd

import cwr_oracle as oracle
from util import GOLDEN, load_plan


def test_host_part_of_coefficient_derivation_matches_oracle():
    from clearwater_riverine_amd.model import face_to_face_distance, change_in_time
    mesh, _, _ = load_plan('plan01', 0.01)
    assert np.array_equal(face_to_face_distance(mesh), mesh['face_to_face_dist'])
    dt = change_in_time(mesh['time_seconds'])
    assert np.array_equal(dt[:-1], mesh['dt'][:-1]) and np.isnan(dt[-1])
    stamps = np.array(['2023-01-01T12:00:00', '2023-01-01T12:05:00', '2023-01-01T12:10:00'], dtype='datetime64[ns]')
    assert np.array_equal(change_in_time(stamps)[:2], [300.0, 300.0])


def test_input_array_from_the_reference_csv_data(tmp_path):
    """constituents.py:78-164 on the reference's own CSV data (plan02): IC row 0, BC in ghost cells 4 and 6."""
    from clearwater_riverine_amd.model import Mesh, input_array_from_csv
    z = np.load(os.path.join(GOLDEN, 'plan02_inputs.npz'))
    ic = tmp_path / 'ic.csv'
    bc = tmp_path / 'bc.csv'
    pd.DataFrame({'Cell_Index': z['ic_cell_index'], 'Concentration': z['ic_concentration']}).to_csv(ic, index=False)
    pd.DataFrame({'RAS2D_TS_Name': z['bc_csv_name'], 'Datetime': z['bc_csv_datetime'],
                  'Concentration': z['bc_csv_concentration']}).to_csv(bc, index=False)
    secs = oracle.parse_ras_stamps(z['time_stamps'])
    time = np.datetime64('2023-01-01T12:00:00') + (secs * 1e9).astype('timedelta64[ns]')
    mesh = Mesh({'time': time, 'face_x': z['face_x'], 'edges_face2': z['edges_face2']})
    faces = {str(nm): [int(f) for f, l in zip(z['bc_face_index'], z['bc_face_line']) if l == i]
             for i, nm in enumerate(z['bc_line_names'])}
    arr = input_array_from_csv(mesh, str(ic), str(bc), faces)
    assert arr.shape == (25, 8)
    assert np.all(arr[0, :2] == 100.0)
    assert np.all(arr[:, 4] == 100.0) and np.all(arr[:, 6] == 100.0)
    assert np.all(arr[1:, [0, 1, 2, 3, 5, 7]] == 0.0)
    _, want, _ = load_plan('plan02', 0.01)
    assert np.array_equal(arr, want)


def test_vectorised_boundary_pipeline_equals_the_reference_algorithm_and_is_fast():
    """constituents.py:100-164 (merge_asof per line, interpolate, concat, left merge, fancy assignment)
    restated literally in the oracle vs the product's vectorised pipeline, on a case of the Ohio River model's size
    (913 hourly stamps, 3 boundary lines with 40 / 25 / 60 faces; the reference needs 12 s there, Ohio River.ipynb
    cell[13]): CSV rows at irregular times, gaps (interpolated), rows between stamps, unsorted rows.  (Every line starts
    at or before the first model stamp: otherwise merge_asof leaves the line NAME empty on the leading stamps and the
    reference's left merge + fancy index raises; the vectorised pipeline writes NaN there.)"""
    import time
    from clearwater_riverine_amd.model import Mesh, input_array_from_csv
    rng = np.random.default_rng(5)
    T, nreal, E = 913, 3000, 6500
    t0 = np.datetime64('2010-05-29T00:00:00')
    stamps = t0 + (np.arange(T) * 3600).astype('timedelta64[s]')
    f2 = rng.integers(0, nreal, size=E)
    lines = {'Upstream Q': np.arange(100, 140), 'Tributary': np.arange(900, 925), 'Downstream Stage': np.arange(5000, 5060)}
    ghost = nreal
    for faces in lines.values():
        f2[faces] = ghost + np.arange(len(faces))
        ghost += len(faces)
    mesh = Mesh({'time': stamps.astype('datetime64[ns]'), 'face_x': np.zeros(ghost), 'edges_face2': f2})
    rows = []
    for name, start, every in (('Upstream Q', -5, 7200), ('Tributary', -30, 5400), ('Downstream Stage', 0, 86400)):
        when = t0 + (start * 3600 + np.sort(rng.choice(np.arange(0, T * 3600, every), size=min(120, len(np.arange(0, T * 3600, every))), replace=False))).astype('timedelta64[s]')
        when[0] = t0 + np.timedelta64(min(start, 0) * 3600, 's')                       # a row at or before the first stamp
        conc = 50.0 + 40.0 * rng.random(len(when))
        conc[1 + rng.choice(len(when) - 1, size=len(when) // 10, replace=False)] = np.nan    # gaps in the series
        rows.append(pd.DataFrame({'RAS2D_TS_Name': name, 'Datetime': when, 'Concentration': conc}))
    bc_df = pd.concat(rows, ignore_index=True).sample(frac=1.0, random_state=1).reset_index(drop=True)      # unsorted rows
    boundary_data = pd.DataFrame([{'BC Line ID': i, 'Face Index': int(f), 'Name': nm, 'Type': 'External'}
                                  for i, (nm, faces) in enumerate(lines.items()) for f in faces])
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        icp, bcp = os.path.join(d, 'ic.csv'), os.path.join(d, 'bc.csv')
        pd.DataFrame({'Cell_Index': np.arange(nreal), 'Concentration': 1.0 + rng.random(nreal)}).to_csv(icp, index=False)
        bc_df.to_csv(bcp, index=False)
        w0 = time.perf_counter()
        got = input_array_from_csv(mesh, icp, bcp, boundary_data)
        fast = time.perf_counter() - w0
        got_dict = input_array_from_csv(mesh, icp, bcp, {k: v.tolist() for k, v in lines.items()})
        ic = pd.read_csv(icp)
        bc_read = pd.read_csv(bcp, parse_dates=['Datetime'])
    want = np.zeros((T, ghost))
    want[0, ic['Cell_Index'].to_numpy()] = ic['Concentration'].to_numpy()
    w0 = time.perf_counter()
    # merge_asof needs each group sorted on the key; the reference's CSVs are (its groupby keeps row order), ours is shuffled
    oracle.set_boundary_conditions_literal(want, mesh, bc_read.sort_values('Datetime', kind='stable'), boundary_data)
    slow = time.perf_counter() - w0
    assert np.array_equal(got, want, equal_nan=True) and np.array_equal(got_dict, want, equal_nan=True)
    assert np.isfinite(got[-1, nreal:]).all()
    assert fast < 1.0, f'vectorised boundary pipeline took {fast:.2f} s (literal restatement of the reference: {slow:.2f} s)'


def test_boundary_dataframe_has_the_reference_columns():
    """hdf_reader.boundary_dataframe == io/hdf.py:355-436 on records shaped like the HDF's compound datasets."""
    from clearwater_riverine_amd.hdf_reader import boundary_dataframe
    from clearwater_riverine_amd.mass_balance import boundary_lines
    ext = np.array([(0, 10, 0, 1, 0.0, 5.0), (0, 11, 1, 2, 5.0, 9.0), (1, 40, 0, 1, 0.0, 3.0), (1, 41, 1, 2, 3.0, 6.0), (1, 41, 1, 2, 3.0, 6.0)],
                   dtype=[('BC Line ID', '<i4'), ('Face Index', '<i4'), ('FP Start Index', '<i4'), ('FP End Index', '<i4'),
                          ('Station Start', '<f4'), ('Station End', '<f4')])
    att = np.array([(b'Upstream Q', b'Perimeter 1', b'External', 9.0), (b'Downstream', b'Perimeter 1', b'External', 6.0)],
                   dtype=[('Name', 'O'), ('SA-2D', 'O'), ('Type', 'O'), ('Length', '<f4')])
    keep = np.array([True, True, True, False, False])               # face 41 is not in the line's 'Faces' attribute
    df = boundary_dataframe(ext, att, keep)
    assert list(df.columns) == ['BC Line ID', 'Face Index', 'FP Start Index', 'FP End Index', 'Name', 'SA-2D', 'Type', 'Length']
    assert df['Face Index'].tolist() == [10, 11, 40] and df['Name'].tolist() == ['Upstream Q', 'Upstream Q', 'Downstream']
    assert [(n, f.tolist()) for n, f in boundary_lines(df)] == [('Upstream Q', [10, 11]), ('Downstream', [40])]


def test_synthetic_mesh_has_the_reference_surface_and_discrete_continuity():
    import clearwater_riverine_amd as cw
    m = cw.synthetic.make_mesh(30, 12, 10, seed=2, n_merge=40, n_dry=3)
    n = m['nreal'] + 1
    f1, f2 = m['edges_face1'], m['edges_face2']
    assert f1.dtype == np.int32 and m['face_flow'].dtype == np.float32 and m['volume'].dtype == np.float32
    assert f1.max() == m['nreal'] and f1.min() >= 0                     # face1 always real (io/hdf.py:268)
    ghosts = f2[f2 > m['nreal']]
    assert len(np.unique(ghosts)) == len(ghosts)                        # one ghost cell per perimeter face
    assert len(m['face_x']) == n + len(ghosts)
    assert n == 30 * 12 - 40
    deg = np.bincount(np.concatenate([f1, f2[f2 <= m['nreal']]]), minlength=n)
    assert set(np.unique(deg)) >= {4, 6}                                # mixed cell degrees
    assert np.count_nonzero(m['volume'][0, :n] == 0) == 3               # the dry cells (dummy-diagonal rows)
    # discrete continuity of the float32 field to float32 rounding (mesh without dry cells)
    m = cw.synthetic.make_mesh(30, 12, 10, seed=2, n_merge=40)
    f1, f2 = m['edges_face1'], m['edges_face2']
    dt = np.diff(m['time_seconds'])[0]
    wet = m['volume'][0, :n] > 0
    for t in (0, 5):
        a = m['face_flow'][t].astype(np.float64)
        div = np.bincount(f1, weights=a, minlength=n)
        internal = f2 <= m['nreal']
        div -= np.bincount(f2[internal], weights=a[internal], minlength=n)
        dv = m['volume'][t + 1, :n].astype(np.float64) - m['volume'][t, :n].astype(np.float64)
        assert np.max(np.abs(dv + dt * div)[wet]) <= 2e-4 * np.max(m['volume'][t, :n])
    assert np.all(m['edge_velocity'][m['face_flow'] == 0] == 0)         # walls: zero flow, zero velocity


def test_boundary_input_array_convention():
    import clearwater_riverine_amd as cw
    m = cw.synthetic.make_mesh(10, 6, 4, seed=0)
    arr = cw.synthetic.boundary_input_array(m, 3)
    n = m['nreal'] + 1
    assert arr.shape == (5, len(m['face_x']), 3)
    assert np.all(arr[0, :n, 1] == 2.0) and np.all(arr[1:, :n] == 0.0)
    assert np.all(arr[:, m['inlet_ghost_cells'], 0] > 0)
    assert np.all(arr[:, m['wall_ghost_cells']] == 0)                    # zero = "no boundary value"


def test_mesh_attribute_surface():
    from clearwater_riverine_amd.model import Mesh
    m = Mesh({'volume': np.zeros(3)})
    m.attrs['nreal'] = 2
    m.attrs['diffusion_coefficient'] = 0.1
    assert m.nreal == 2 and m.diffusion_coefficient == 0.1 and m.volume.shape == (3,)


def test_hilbert_order_is_a_permutation_with_local_neighbours():
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.ordering import hilbert_order, renumber_mesh
    m = cw.synthetic.make_mesh(64, 48, 2, seed=3, n_merge=50)
    n = m['nreal'] + 1
    order = hilbert_order(m['face_x'], m['face_y'], n)
    assert np.array_equal(np.sort(order), np.arange(n))
    r = renumber_mesh(m, order)
    assert np.array_equal(r['face_x'][:n], m['face_x'][order]) and np.array_equal(r['volume'][:, :n], m['volume'][:, order])
    assert np.array_equal(r['face_x'][n:], m['face_x'][n:])             # ghosts untouched
    # same faces between the same cells, expressed in the new ids
    f1, f2 = np.asarray(m['edges_face1']), np.asarray(m['edges_face2'])
    full = np.arange(len(m['face_x'])); full[:n] = order
    assert np.array_equal(full[r['edges_face1']], f1) and np.array_equal(full[r['edges_face2']], f2)
    # locality: the median id distance of face neighbours shrinks well below the grid width
    internal = f2 < n
    d_new = np.abs(r['edges_face1'][internal].astype(np.int64) - r['edges_face2'][internal])
    assert np.median(d_new) <= 8


def test_tile_balanced_numbering_is_a_permutation_that_keeps_the_tiles():
    """ordering.balance_windows: within every tile-sized window of the curve the cells are sorted by their J^2 row length; each
    window keeps exactly its cells (so a tile's LDS image is unchanged), the mean of the per-wave maximum drops, and a
    window <= 1 leaves the order alone."""
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.ordering import balance_windows, hilbert_order, two_hop_row_lengths
    m = cw.synthetic.make_mesh(96, 64, 2, seed=3, n_merge=300)
    n = m['nreal'] + 1
    f1, f2 = m['edges_face1'], m['edges_face2']
    order = hilbert_order(m['face_x'], m['face_y'], n)
    length = two_hop_row_lengths(f1, f2, n)
    assert length.min() >= 4 and length.max() >= 15 and np.bincount(length).argmax() in (9, 10)
    bal = balance_windows(order, f1, f2, window=64)
    assert np.array_equal(np.sort(bal), np.arange(n))
    for w0 in range(0, n, 64):
        assert set(bal[w0:w0 + 64]) == set(order[w0:w0 + 64])
        assert np.all(np.diff(length[bal[w0:w0 + 64]]) >= 0)
    waves = lambda o: length[o][: n // 16 * 16].reshape(-1, 16).max(axis=1).mean()
    assert waves(bal) < waves(order) - 1.0
    assert np.array_equal(balance_windows(order, f1, f2, window=0), order)


def test_synthetic_meshes_with_eight_sided_cells():
    """synthetic.make_mesh(n_merge4=...): 2 x 2 blocks of quads become 8-sided cells (HEC-RAS's maximum), pair merges avoid
    them, and meshes built without the option are what they were before it existed (their random stream is untouched)."""
    import clearwater_riverine_amd as cw
    m = cw.synthetic.make_mesh(40, 30, 3, seed=9, n_merge=60, n_merge4=50)
    n = m['nreal'] + 1
    assert n == 40 * 30 - 60 - 3 * 50
    deg = np.bincount(np.concatenate([m['edges_face1'], m['edges_face2']]), minlength=len(m['face_x']))[:n]
    assert np.array_equal(np.bincount(deg, minlength=9)[[4, 6, 8]], [n - 110, 60, 50]) and deg.max() == 8
    # every face has a real cell on side 1, its ghost (or a different real cell) on side 2
    assert (m['edges_face1'] <= m['nreal']).all() and (m['edges_face1'] != m['edges_face2']).all()
    a = cw.synthetic.make_mesh(24, 10, 3, seed=9, n_merge=12, n_dry=1)
    b = cw.synthetic.make_mesh(24, 10, 3, seed=9, n_merge=12, n_dry=1, n_merge4=0)
    assert all(np.array_equal(a[k], b[k], equal_nan=True) for k in a if isinstance(a[k], np.ndarray))
    assert int(np.asarray(a['edges_face1']).sum()) == 57280      # pinned: the generator's stream for existing seeds


def test_curve_kind_follows_the_size_of_a_rank_and_the_environment(monkeypatch):
    """distributed.curve_kind: lanes along the flow for engines that will chain their tiles (one GPU; ranks of at least 1.75
    tiles per resident block), the isotropic Hilbert curve below that and whenever the chains are switched off."""
    from clearwater_riverine_amd.distributed import curve_kind
    from clearwater_riverine_amd.engine import tile_rows
    for v in ('CWR_TILE_ORDER', 'CWR_NO_CHAINS'):
        monkeypatch.delenv(v, raising=False)
    tr = tile_rows(16)
    assert curve_kind(1_000_000, 16, 1) == 'lanes' and curve_kind(1_000_000, 1, 1) == 'lanes'
    assert curve_kind(100_000, 16, 1) == 'hilbert' and curve_kind(10_000, 12, 1) == 'hilbert'      # one GPU below the chain threshold (round 4)
    assert curve_kind(1_000_000, 16, 2) == 'lanes' and curve_kind(1_000_000, 16, 4) == 'lanes'
    # 125 k cells per rank: 1.9 tiles per block -- chained since the lane boundaries were smoothed (1.75 tiles per block; was 3)
    assert curve_kind(1_000_000, 16, 8) == 'lanes' and curve_kind(120_000, 16, 1) == 'lanes'
    assert curve_kind(1_000_000, 1, 2) == 'lanes' and curve_kind(1_000_000, 1, 4) == 'hilbert'       # (256-row tiles at K = 1)
    from clearwater_riverine_amd.engine import chain_min_rows
    lim = chain_min_rows(16)
    assert lim == int(1.75 * 1024) * tr                          # (four resident blocks on each of 256 CUs; no GPU here: the default)
    assert curve_kind(lim * 2, 16, 2) == 'lanes' and curve_kind(lim * 2 - 2, 16, 2) == 'hilbert'
    # the numbering follows the ENGINE's threshold: CWR_CHAIN_MIN_TILES moves both (round 5; it used to move the engine's only)
    monkeypatch.setenv('CWR_CHAIN_MIN_TILES', '3')
    assert chain_min_rows(16) == 3 * 1024 * tr and curve_kind(120_000, 16, 1) == 'hilbert' and curve_kind(250_000, 16, 1) == 'lanes'
    monkeypatch.delenv('CWR_CHAIN_MIN_TILES')
    monkeypatch.setenv('CWR_NO_CHAINS', '1')
    assert curve_kind(1_000_000, 16, 1) == 'hilbert'
    monkeypatch.setenv('CWR_TILE_ORDER', 'lanes')
    assert curve_kind(1_000_000, 16, 8) == 'lanes'
    monkeypatch.setenv('CWR_TILE_ORDER', 'hilbert')
    monkeypatch.delenv('CWR_NO_CHAINS')
    assert curve_kind(1_000_000, 16, 1) == 'hilbert'


def _tile_extent(order, mesh, n, tile=64, dx=10.0):
    """Median (along, across) extent in cells of the tiles of `order`, measured in the coordinates of the UNBENT mesh."""
    x = np.asarray(mesh['face_x'])[:n][order] / dx
    y = np.asarray(mesh['face_y'])[:n][order] / dx
    nt = n // tile
    xs = x[:nt * tile].reshape(nt, tile); ys = y[:nt * tile].reshape(nt, tile)
    return float(np.median(xs.max(1) - xs.min(1))), float(np.median(ys.max(1) - ys.min(1)))


def test_lanes_follow_a_channel_that_bends(monkeypatch):
    """ordering.lane_order on a meander (synthetic.bend_channel: same cells, faces and flows, laid along a sine-generated centre
    line).  Straight lanes along the principal axis cut across the bends; the curvilinear coordinates (distance from the longest
    bank through the mesh, cross-sections perpendicular to the banks) give lanes that are stream tubes again and tiles as compact
    as on the straight channel.  A straight channel keeps its straight lanes."""
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd import ordering as od
    monkeypatch.delenv('CWR_LANE_KIND', raising=False)
    monkeypatch.delenv('CWR_LANE_LEN', raising=False)
    mesh = cw.synthetic.make_mesh(384, 96, 6, seed=4, dt=40.0, diffusion_coefficient=0.5, n_merge=1800)
    n = mesh['nreal'] + 1
    bent = cw.synthetic.bend_channel(mesh, 1.0)
    # (what moved: coordinates only)
    assert np.array_equal(bent['edges_face1'], mesh['edges_face1']) and bent['face_flow'] is mesh['face_flow']
    assert not np.allclose(bent['face_y'][:n], mesh['face_y'][:n])
    sig, q = od.channel_coordinates(bent, n)
    assert np.all(np.isfinite(sig)) and np.all(np.isfinite(q)) and q.min() == 0.0
    # q is the distance from one bank, sigma runs along the channel: in the unbent coordinates they are y and x again
    y0 = np.asarray(mesh['face_y'])[:n]; x0 = np.asarray(mesh['face_x'])[:n]
    cy = abs(np.corrcoef(q, y0)[0, 1]); cx = abs(np.corrcoef(sig, x0)[0, 1])
    # (sigma counts along the reference BANK, which a bend makes shorter or longer than the centre line: monotone, not linear)
    assert cy > 0.995 and cx > 0.98, (cy, cx)
    width = 21 * 10.0                                            # straight lanes: 21 cells wide (tiles 3 long); curvilinear: 16 (4 long)
    lanes_c = np.floor(q / (16 * 10.0)).astype(np.int64)
    (ax, ay), ratio = od.flow_axis(bent, n)
    qa = -np.asarray(bent['face_x'])[:n] * ay + np.asarray(bent['face_y'])[:n] * ax
    lanes_s = np.floor((qa - qa.min()) / width).astype(np.int64)
    c_cur, c_str = od.cross_lane_flow(lanes_c, bent, n), od.cross_lane_flow(lanes_s, bent, n)
    assert c_cur < 0.012 and c_str > 2.5 * c_cur, (c_cur, c_str)
    order = od.lane_order(bent, n)
    assert np.array_equal(np.sort(order), np.arange(n))
    monkeypatch.setenv('CWR_LANE_KIND', 'straight')
    order_s = od.lane_order(bent, n)
    monkeypatch.delenv('CWR_LANE_KIND')
    assert not np.array_equal(order, order_s)                     # the curvilinear lanes were chosen ...
    along, across = _tile_extent(order, mesh, n)
    along_s, across_s = _tile_extent(order_s, mesh, n)
    assert along <= 5.5 and across <= 16.5 and along_s >= 2 * along, (along, across, along_s, across_s)   # ... and their tiles are 4-5 x 16 cells
    # the straight channel: its straight lanes are stream tubes already and stay (the bench mesh's numbering is what it was)
    order0 = od.lane_order(mesh, n)
    monkeypatch.setenv('CWR_LANE_KIND', 'straight')
    assert np.array_equal(order0, od.lane_order(mesh, n))
    monkeypatch.setenv('CWR_LANE_KIND', 'channel')
    along_c, across_c = _tile_extent(od.lane_order(mesh, n), mesh, n)
    assert along_c <= 5.5 and across_c <= 16.5
    # no inflow boundary (every perimeter face closed) -> no channel coordinates; a field without an axis then keeps the Hilbert curve
    closed = dict(mesh); closed['face_flow'] = np.zeros_like(mesh['face_flow'])
    assert od.channel_coordinates(closed, n) is None
    monkeypatch.delenv('CWR_LANE_KIND')
    assert np.array_equal(od.lane_order(closed, n), od.hilbert_order(closed['face_x'], closed['face_y'], n))
    # a NaN in the flow field (a face HEC-RAS left undefined) is neither an open boundary nor a bank: the order is still a permutation
    holed = dict(bent); ff = np.array(bent['face_flow'], dtype=np.float32, copy=True); ff[:, ::97] = np.nan; holed['face_flow'] = ff
    assert np.array_equal(np.sort(od.lane_order(holed, n)), np.arange(n))
    # a reversing (tidal) field: the net flow through the open boundaries over the sampled levels decides which end is the inflow;
    # when it cancels, the end with the largest flow serves (the arc length only needs AN end to start from): same lanes, same tiles
    tidal = dict(bent); ft = np.array(bent['face_flow'], dtype=np.float32, copy=True); ft[1::2] *= -1.0
    tidal['face_flow'] = ft[:6]                                    # (an even number of levels: sampled net flow exactly zero)
    assert od.channel_coordinates(tidal, n) is not None
    order_t = od.lane_order(tidal, n)
    assert np.array_equal(np.sort(order_t), np.arange(n))
    along_t, across_t = _tile_extent(order_t, mesh, n)
    assert along_t <= 5.5 and across_t <= 16.5, (along_t, across_t)
