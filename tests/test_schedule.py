"""CPU tests of the host side of the chained passes: the numpy specification of the engine's tile chains (schedule.py, compared
with the engine's own builder in tests/test_gpu_chains.py) and the lane-major cell order (ordering.lane_order)."""
import numpy as np

import clearwater_riverine_amd as cw
from clearwater_riverine_amd import schedule as sch
from clearwater_riverine_amd.ordering import flow_axis, hilbert_order, lane_order, renumber_mesh


def _mesh(nx=96, ny=64, **kw):
    return cw.synthetic.make_mesh(nx, ny, 3, seed=5, n_merge=nx * ny // 20, dt=40.0, diffusion_coefficient=0.5, **kw)


def test_flow_axis_and_lane_order_follow_the_flow():
    mesh = _mesh()
    n = mesh['nreal'] + 1
    (ax, ay), ratio = flow_axis(mesh, n)
    assert abs(abs(ax) - 1.0) < 0.05 and abs(ay) < 0.3 and ratio > 1.5          # the synthetic through-flow runs along x
    order = lane_order(mesh, n, tile_rows=64)
    assert np.array_equal(np.sort(order), np.arange(n))                           # a permutation of the real cells
    x, y = np.asarray(mesh['face_x'])[order], np.asarray(mesh['face_y'])[order]
    ext_x = np.array([np.ptp(x[i:i + 64]) for i in range(0, n - 63, 64)])
    ext_y = np.array([np.ptp(y[i:i + 64]) for i in range(0, n - 63, 64)])
    # a tile (64 consecutive cells) is short along the flow and a lane wide: ~3 x 21 cells of 10 m (the few tiles that straddle
    # two lanes excepted)
    assert np.median(ext_x) < 0.5 * np.median(ext_y)
    assert 20.0 <= np.median(ext_x) <= 50.0 and 160.0 <= np.median(ext_y) <= 250.0
    # consecutive tiles follow each other along the flow: neighbouring tile centres are about one tile length apart
    cx = np.array([x[i:i + 64].mean() for i in range(0, n - 63, 64)]); cy = np.array([y[i:i + 64].mean() for i in range(0, n - 63, 64)])
    step = np.hypot(np.diff(cx), np.diff(cy))
    assert np.median(step) < 60.0


def test_field_without_a_preferred_axis_keeps_the_hilbert_curve():
    mesh = _mesh(48, 48)
    n = mesh['nreal'] + 1
    rng = np.random.default_rng(0)
    mesh['face_flow'] = rng.standard_normal(mesh['face_flow'].shape).astype(np.float32)      # isotropic noise
    assert flow_axis(mesh, n)[1] < 1.5
    assert np.array_equal(lane_order(mesh, n, tile_rows=64), hilbert_order(mesh['face_x'], mesh['face_y'], n))


def test_chain_schedule_covers_every_tile_once_and_follows_the_flow():
    mesh = _mesh()
    n = mesh['nreal'] + 1
    m = renumber_mesh(mesh, lane_order(mesh, n, tile_rows=64))
    TR, grid = 64, 16
    ntiles = -(-n // TR)
    us, ud, w = sch.tile_links(m['edges_face1'], m['edges_face2'], m['face_flow'][0], n, TR, ntiles)
    assert (us != ud).all() and (w > 0).all()
    chains = sch.chains(us, ud, w, ntiles)
    flat = np.concatenate([np.asarray(c) for c in chains])
    assert np.array_equal(np.sort(flat), np.arange(ntiles))
    assert max(len(c) for c in chains) >= 8                       # lanes: whole runs of tiles are linked
    # every link of a chain carries the largest outflow of its tile
    best = {}
    for a, b, f in zip(us, ud, w):
        if f > best.get(a, (0.0, -1))[0]:
            best[a] = (f, b)
    for c in chains:
        for a, b in zip(c[:-1], c[1:]):
            assert best[a][1] == b
    for spb in (1, 2):
        s = sch.schedule(chains, ntiles, grid, streams_per_block=spb)
        assert s.shape[1] == grid
        tiles = s[s >= 0]
        assert np.array_equal(np.sort(tiles), np.arange(ntiles))
        assert (np.diff((s >= 0).astype(int), axis=0) <= 0).all()  # dense prefixes, -1 padded
        assert np.ptp((s >= 0).sum(axis=0)) <= spb                 # balanced lists
    # a reversed field reverses the chains' direction, not their membership
    rev = sch.chains(*sch.tile_links(m['edges_face1'], m['edges_face2'], -m['face_flow'][0], n, TR, ntiles), ntiles)
    assert sorted(len(c) for c in rev) == sorted(len(c) for c in chains)
