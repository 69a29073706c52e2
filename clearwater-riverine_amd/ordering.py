"""Internal cell renumbering (host side, numpy): a space-filling-curve order of the real cells.

The reference's cell ids (HEC-RAS order) stay the ids of every boundary array; the engine may work in
a permuted numbering in which a cell's face neighbours -- and the neighbours' neighbours the J^2 pass
gathers -- are close in memory (SURVEY.md section 7 "gather locality").  Row sums do not depend on the
numbering (the faces of a row are always visited in ascending face id), so concentrations are bitwise
identical with and without renumbering; only the order in which norms are accumulated changes.
Ghost cells are never renumbered.
"""
from __future__ import annotations

import numpy as np


def hilbert_index(gx: np.ndarray, gy: np.ndarray, bits: int = 16) -> np.ndarray:
    """Hilbert-curve index of integer grid points (vectorised xy -> d)."""
    x = gx.astype(np.int64).copy()
    y = gy.astype(np.int64).copy()
    d = np.zeros_like(x)
    s = 1 << (bits - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        swap = ry == 0
        flip = swap & (rx == 1)
        x = np.where(flip, s - 1 - x, x)
        y = np.where(flip, s - 1 - y, y)
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s >>= 1
    return d


def hilbert_order(face_x, face_y, n_real: int) -> np.ndarray:
    """order[new id] = reference id, for the real cells 0..n_real-1, along a Hilbert curve through the cell centres."""
    x = np.asarray(face_x, dtype=np.float64)[:n_real]
    y = np.asarray(face_y, dtype=np.float64)[:n_real]
    span = max(float(x.max() - x.min()), float(y.max() - y.min()), 1e-300)
    gx = np.minimum(((x - x.min()) / span * 65535.0), 65535.0).astype(np.int64)
    gy = np.minimum(((y - y.min()) / span * 65535.0), 65535.0).astype(np.int64)
    return np.argsort(hilbert_index(gx, gy), kind='stable').astype(np.int64)


def renumber_mesh(mesh: dict, order: np.ndarray) -> dict:
    """A shallow copy of `mesh` in the numbering order[new] = old (real cells only; ghost ids unchanged)."""
    n = len(order)
    ncell = len(mesh['face_x'])
    inv = np.arange(ncell, dtype=np.int64)
    inv[order] = np.arange(n)
    full = np.arange(ncell, dtype=np.int64)
    full[:n] = order
    m = dict(mesh)
    m['edges_face1'] = inv[np.asarray(mesh['edges_face1'])].astype(np.int32)
    m['edges_face2'] = inv[np.asarray(mesh['edges_face2'])].astype(np.int32)
    m['face_x'] = np.asarray(mesh['face_x'])[full]
    m['face_y'] = np.asarray(mesh['face_y'])[full]
    m['volume'] = np.ascontiguousarray(np.asarray(mesh['volume'])[:, full])
    return m


def two_hop_row_lengths(face1, face2, n_real: int) -> np.ndarray:
    """Entries per row of the squared Jacobi operator J^2 (columns reachable in two face steps; the engine's
    ensure_sq_pattern builds exactly this pattern): the work of a row in the tiled pass."""
    try:
        from scipy.sparse import csr_matrix
    except ImportError:                                   # no scipy: every row counts the same, the order stays as it is
        return np.zeros(n_real, dtype=np.int64)
    f1 = np.asarray(face1, dtype=np.int64)
    f2 = np.asarray(face2, dtype=np.int64)
    real = (f1 < n_real) & (f2 < n_real)
    a, b = f1[real], f2[real]
    A = csr_matrix((np.ones(2 * len(a), dtype=np.int8), (np.concatenate([a, b]), np.concatenate([b, a]))), shape=(n_real, n_real))
    A.data[:] = 1
    A2 = (A.astype(np.int32) @ A.astype(np.int32)).tocsr()
    return np.diff(A2.indptr).astype(np.int64)


def balance_windows(order: np.ndarray, face1, face2, window: int = 64) -> np.ndarray:
    """Within every `window` consecutive positions of `order` (one tile of the tiled pass, or a whole number of its waves),
    sort the cells by their J^2 row length.  A wave of the tiled pass loops to the longest of its rows; on an unstructured
    mesh the few 5-8-face cells (18-40 entries among rows of 9-10) otherwise sit in almost every wave.  The tile still holds
    the same cells (all of them in LDS), so locality is unchanged; results do not depend on the numbering."""
    order = np.asarray(order, dtype=np.int64)
    n = len(order)
    if window <= 1:
        return order
    length = two_hop_row_lengths(face1, face2, n)[order]
    win = np.arange(n) // window
    return order[np.lexsort((np.arange(n), length, win))]


def flow_axis(mesh: dict, n_real: int, max_levels: int = 8):
    """Principal axis of the flow field: unit vector (ax, ay) maximising sum over (sampled) levels and internal faces of
    (face_flow * cos(angle between the axis and the face1 -> face2 direction))^2, and the ratio of the two eigenvalues of that
    2 x 2 moment matrix (1 = no preferred direction).  Squared flows: a reversing (tidal) field keeps its axis."""
    f1 = np.asarray(mesh['edges_face1'], dtype=np.int64)
    f2 = np.asarray(mesh['edges_face2'], dtype=np.int64)
    x = np.asarray(mesh['face_x'], dtype=np.float64)
    y = np.asarray(mesh['face_y'], dtype=np.float64)
    real = f2 < n_real
    dx, dy = (x[f2] - x[f1])[real], (y[f2] - y[f1])[real]
    ln = np.maximum(np.hypot(dx, dy), 1e-300)
    dx, dy = dx / ln, dy / ln
    flow = np.asarray(mesh['face_flow'])
    T = flow.shape[0]
    levels = np.unique(np.linspace(0, T - 1, min(T, max_levels)).astype(np.int64))
    q2 = np.zeros(int(real.sum()))
    for t in levels:
        q2 += np.asarray(flow[t], dtype=np.float64)[real] ** 2
    M = np.array([[np.sum(q2 * dx * dx), np.sum(q2 * dx * dy)], [np.sum(q2 * dx * dy), np.sum(q2 * dy * dy)]])
    if not np.all(np.isfinite(M)) or M.trace() <= 0:
        return (1.0, 0.0), 1.0
    w, v = np.linalg.eigh(M)
    return (float(v[0, 1]), float(v[1, 1])), float(w[1] / max(w[0], 1e-300 * w[1]))


def _cell_adjacency(a: np.ndarray, b: np.ndarray, n_real: int):
    """0/1 adjacency of the real cells (one entry per face, both directions) and the cells' face counts, for neighbour averaging."""
    from scipy.sparse import csr_matrix
    adj = csr_matrix((np.ones(2 * len(a)), (np.concatenate([a, b]), np.concatenate([b, a]))), shape=(n_real, n_real))
    return adj, np.maximum(np.asarray(adj.sum(axis=1)).ravel(), 1.0)


def _sample_levels(T: int, max_levels: int = 8) -> np.ndarray:
    return np.unique(np.linspace(0, T - 1, min(T, max_levels)).astype(np.int64))


def channel_coordinates(mesh: dict, n_real: int, max_levels: int = 8, smooth: int = 64):
    """Curvilinear coordinates of a channel that bends: q = distance from the longest bank, measured THROUGH the mesh (shortest
    paths over the cell graph, edge = centre-to-centre distance, then smoothed), and sigma = arc length along that bank of the
    point a cell reaches by walking down q -- cross-sections perpendicular to the banks, counted along the bank -- so that lanes
    of constant q and their cells in order of sigma follow a meander where a straight axis cannot.  Open boundary = a perimeter face that carries flow at a sampled level (inflow:
    net flow into the real cell); bank = real cells at perimeter faces that never do; the banks fall into connected pieces along
    the perimeter, the one with the most cells is the reference bank.  Returns (sigma, q) or None (no inflow, no bank, or cells that
    cannot be reached)."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components, dijkstra
    f1 = np.asarray(mesh['edges_face1'], dtype=np.int64)
    f2 = np.asarray(mesh['edges_face2'], dtype=np.int64)
    x = np.asarray(mesh['face_x'], dtype=np.float64)
    y = np.asarray(mesh['face_y'], dtype=np.float64)
    flow = np.asarray(mesh['face_flow'])
    real = (f1 < n_real) & (f2 < n_real)
    peri = ~real
    if not real.any() or not peri.any():
        return None
    levels = _sample_levels(flow.shape[0], max_levels)
    net = np.zeros(len(f1)); mag = np.zeros(len(f1))
    for t in levels:
        ft = np.asarray(flow[t], dtype=np.float64)
        net += ft; mag = np.maximum(mag, np.abs(ft))
    inner = np.where(f1 < n_real, f1, f2)                    # the real cell of a perimeter face (face1, in HEC-RAS output)
    into = np.where(f1 < n_real, net < 0, net > 0)            # flow(face1 -> face2) < 0 enters face1
    inflow = np.unique(inner[peri & (mag > 0) & into])
    bank = np.unique(inner[peri & (mag == 0)])
    if len(inflow) == 0 and np.any(peri & (mag > 0)):
        # a reversing (tidal) field whose sampled net flow cancels: either end will do -- the distance from it only says which end of
        # the reference bank the arc length starts from; the engine's chains follow the flow of the level being solved
        inflow = np.array([inner[np.argmax(np.where(peri, mag, -1.0))]])
    if len(inflow) == 0 or len(bank) == 0:
        return None
    a, b = f1[real], f2[real]
    w = np.maximum(np.hypot(x[a] - x[b], y[a] - y[b]), 1e-300)
    g = csr_matrix((np.concatenate([w, w]), (np.concatenate([a, b]), np.concatenate([b, a]))), shape=(n_real, n_real))
    s = dijkstra(g, directed=False, indices=inflow, min_only=True)
    is_bank = np.zeros(n_real, dtype=bool); is_bank[bank] = True
    keep = is_bank[a] & is_bank[b]
    sub = csr_matrix((np.ones(int(keep.sum())), (a[keep], b[keep])), shape=(n_real, n_real))
    _, lab = connected_components(sub, directed=False)
    piece = np.bincount(lab[bank]).argmax()                   # (cells off the banks are singletons of their own)
    q = dijkstra(g, directed=False, indices=bank[lab[bank] == piece], min_only=True)
    if not (np.all(np.isfinite(s)) and np.all(np.isfinite(q))):
        return None
    # shortest paths zig-zag through a jittered mesh, so the contours of q are ragged at the scale of a cell -- and every step of a
    # lane boundary is a face the main flow crosses.  A few sweeps of neighbour averaging (the reference bank held at 0) remove
    # the cell-scale noise and leave the shape: the lanes' cross flow falls by 3-4 x, to that of straight lanes on a straight channel
    adj, deg = _cell_adjacency(a, b, n_real)
    ref = bank[lab[bank] == piece]
    fixed = np.zeros(n_real, dtype=bool); fixed[ref] = True
    for _ in range(smooth):
        q = np.where(fixed, q, 0.5 * q + 0.5 * (adj @ q) / deg)
    # The distance from the inflow boundary is no coordinate ALONG the lanes: shortest paths cut every bend on its inside, so its
    # contours lean across the channel (by the channel's width per radian turned) and 64 consecutive cells of a lane would be a
    # sheared stripe.  Cross-sections perpendicular to the banks instead: sigma = arc length along the reference bank, carried
    # into the channel down the slope of q -- every cell takes the q-weighted mean of its neighbours nearer the bank, which in
    # ascending order of q is one sparse triangular solve.  (A cell with no neighbour nearer the bank -- a dimple that the
    # smoothing left -- keeps its distance from the inflow boundary.)
    from scipy.sparse.linalg import spsolve_triangular
    keep = fixed[a] & fixed[b]
    gb = csr_matrix((np.concatenate([w[keep], w[keep]]), (np.concatenate([a[keep], b[keep]]), np.concatenate([b[keep], a[keep]]))),
                    shape=(n_real, n_real))
    start = ref[np.argmin(s[ref])]
    sig_ref = dijkstra(gb, directed=False, indices=start)
    if not np.all(np.isfinite(sig_ref[ref])):
        return None
    rank = np.empty(n_real, dtype=np.int64)
    perm = np.lexsort((np.arange(n_real), q))                          # ascending q (ties: by id)
    rank[perm] = np.arange(n_real)
    aa = np.concatenate([a, b]); bb = np.concatenate([b, a])
    down = rank[bb] < rank[aa]                                          # bb is nearer the bank than aa
    wgt = np.maximum(q[aa[down]] - q[bb[down]], 1e-12 * max(float(q.max()), 1e-300))
    rows, cols = rank[aa[down]], rank[bb[down]]
    free = ~fixed[aa[down]]
    tot = np.bincount(rows[free], weights=wgt[free], minlength=n_real)
    orphan = (tot == 0) & ~fixed[perm]
    rhs = np.zeros(n_real)
    rhs[rank[ref]] = sig_ref[ref]
    rhs[orphan] = s[perm][orphan]
    m_low = csr_matrix((-wgt[free] / tot[rows[free]], (rows[free], cols[free])), shape=(n_real, n_real))
    from scipy.sparse import identity
    sig = spsolve_triangular((identity(n_real, format='csr') + m_low).tocsr(), rhs, lower=True)
    sigma = np.empty(n_real); sigma[perm] = sig
    return sigma, q


def cross_lane_flow(lane: np.ndarray, mesh: dict, n_real: int, max_levels: int = 8) -> float:
    """Share of the (sampled) flow between real cells that crosses from one lane into another: lanes are stream tubes when it is
    small.  The measure by which lane_order chooses between straight and curvilinear lanes."""
    f1 = np.asarray(mesh['edges_face1'], dtype=np.int64)
    f2 = np.asarray(mesh['edges_face2'], dtype=np.int64)
    real = (f1 < n_real) & (f2 < n_real)
    flow = np.asarray(mesh['face_flow'])
    mag = np.zeros(int(real.sum()))
    for t in _sample_levels(flow.shape[0], max_levels):
        mag += np.abs(np.asarray(flow[t], dtype=np.float64)[real])
    tot = float(mag.sum())
    if tot <= 0:
        return 1.0
    return float(mag[lane[f1[real]] != lane[f2[real]]].sum()) / tot


def lane_order(mesh: dict, n_real: int, tile_rows: int = 64, tile_len: float = 3, min_ratio: float = 1.5) -> np.ndarray:
    """Lane-major order for engines that run chained passes: the cells are cut into LANES -- strips along the principal flow
    axis, tile_rows / tile_len cells wide -- and numbered lane by lane, along the flow inside a lane.  A tile of the engine
    (tile_rows consecutive cells) is then ~tile_len cells long and a lane wide, and CONSECUTIVE tiles follow each other along
    the flow: the engine's flux-built chains become whole lanes, a block of the persistent grid streams down a lane segment,
    the upstream halo of every tile is what the same block has just written (hot in its XCD's L2) and two tile-local J^2
    applications carry information across the whole tile.  Against the Hilbert curve (compact tiles, chains of ~6 tiles whose
    concurrently active tiles are scattered over the XCD's region) -- measured in profiles/r03_c_chained_passes.txt.
    A field without a preferred axis keeps the Hilbert curve.
    tile_len: 4 in round 3; 3 (lanes 21 cells wide at 64-row tiles) since the lane boundaries are smoothed -- same-box pairs on the bench
    workload, ms per step at 4 / 3: K = 16: 2.43 / 2.34, 12: 2.16 / 2.08, 8: 1.77 / 1.66, 4: 1.19 / 1.15, 2: 0.97 / 0.94, 1: 0.81 / 0.79; 2 and 5
    and 8 lose everywhere (profiles/r04_y)."""
    import os
    if os.environ.get('CWR_LANE_LEN'):                       # (A/B knob: tile length along the flow in cells; lanes are tile_rows / it wide)
        tile_len = max(1.0, float(os.environ['CWR_LANE_LEN']))
    (ax, ay), ratio = flow_axis(mesh, n_real)
    x = np.asarray(mesh['face_x'], dtype=np.float64)[:n_real]
    y = np.asarray(mesh['face_y'], dtype=np.float64)[:n_real]
    if n_real < 4 * tile_rows:
        return hilbert_order(mesh['face_x'], mesh['face_y'], n_real)
    f1 = np.asarray(mesh['edges_face1'], dtype=np.int64)
    f2 = np.asarray(mesh['edges_face2'], dtype=np.int64)
    real = f2 < n_real
    h = float(np.median(np.hypot(x[f1[real]] - x[f2[real]], y[f1[real]] - y[f2[real]]))) if real.any() else 1.0
    width = max(1, int(tile_rows / tile_len)) * max(h, 1e-300) * float(os.environ.get('CWR_LANE_WIDTH_SCALE', '1.0'))   # (A/B knob)
    # candidate 1: straight lanes along the principal axis of the flow (a field with no preferred axis has none)
    straight = None
    if ratio >= min_ratio:
        s_along = x * ax + y * ay
        q_across = -x * ay + y * ax
        # The lane boundaries are cut in a SMOOTHED across-coordinate (neighbour averaging over the cell graph): where a boundary
        # runs within a cell's jitter of a row of cell centres, the raw coordinate deals that row's cells out between two lanes,
        # and every such step is a face through which the main flow crosses from one list to another.  Measured on the bench
        # workload, same box (profiles/r04_w): 0 / 4 / 12 / 32 / 64 / 128 sweeps: 2.54 / 2.49 / 2.46 / 2.39-2.46 / 2.45 / 2.47 ms per
        # step (35-37 -> 33-35 sweeps of the solver); the across-coordinate of channel_coordinates is smoothed for the same reason.
        smooth = int(os.environ.get('CWR_LANE_SMOOTH', '32'))
        if smooth > 0:
            adj, deg = _cell_adjacency(f1[real], f2[real], n_real)
            for _ in range(smooth):
                q_across = 0.5 * q_across + 0.5 * (adj @ q_across) / deg
        straight = (s_along, np.floor((q_across - q_across.min()) / width).astype(np.int64))
    # candidate 2 (round 4): lanes that follow the banks of a channel that bends -- s = distance from the inflow boundary, q =
    # distance from the longest bank, both through the mesh.  Taken when its lanes are clearly better stream tubes than the
    # straight ones (less of the flow crosses from lane to lane), or when there is no axis and they are good ones.
    want = os.environ.get('CWR_LANE_KIND', 'auto')                    # auto | straight | channel (A/B knob)
    curved = None
    c_str = cross_lane_flow(straight[1], mesh, n_real) if straight is not None else 1.0
    # (straight lanes that are already stream tubes -- under 1 % of the flow crosses them, 0.65 % on the bench mesh -- are kept
    # without looking further: the two shortest-path passes cost ~2 s per million cells)
    if want == 'channel' or (want == 'auto' and c_str >= 0.01):
        sq = channel_coordinates(mesh, n_real)
        if sq is not None:
            # (curvilinear lanes stay 4 cells long x tile_rows / 4 wide: sigma's cross-sections lean a little against the lanes, which
            # a wider lane turns into longer tiles -- meander of 250 k cells x 16: 0.945 ms per step at 4, 1.114 at 3; profiles/r04_y)
            width_c = max(1, int(tile_rows / max(tile_len, 4.0))) * max(h, 1e-300)
            curved = (sq[0], np.floor(sq[1] / width_c).astype(np.int64))
    pick = straight
    if curved is not None:
        c_cur = cross_lane_flow(curved[1], mesh, n_real)
        if straight is None:
            pick = curved if (c_cur < 0.05 or want == 'channel') else None
        elif want == 'channel' or c_cur < 0.7 * c_str:
            pick = curved
    if pick is None:
        return hilbert_order(mesh['face_x'], mesh['face_y'], n_real)
    s_along, lane = pick
    # snake: odd lanes run against the axis, so that the end of a lane and the start of the next are neighbours (the tile that
    # straddles two lanes stays compact); the engine's chains follow the flow whatever the numbering direction
    key = np.where(lane & 1, -s_along, s_along)
    # (measured and dropped: taking the cells of a lane cross-section by cross-section -- floor(key / h), then across the lane --
    # instead of by their raw along-coordinate: 2.449-2.459 vs 2.439-2.460 ms per step on the bench workload, profiles/r04_w)
    return np.lexsort((key, lane)).astype(np.int64)
