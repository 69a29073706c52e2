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


def lane_order(mesh: dict, n_real: int, tile_rows: int = 64, tile_len: int = 4, min_ratio: float = 1.5) -> np.ndarray:
    """Lane-major order for engines that run chained passes: the cells are cut into LANES -- strips along the principal flow
    axis, tile_rows / tile_len cells wide -- and numbered lane by lane, along the flow inside a lane.  A tile of the engine
    (tile_rows consecutive cells) is then ~tile_len cells long and a lane wide, and CONSECUTIVE tiles follow each other along
    the flow: the engine's flux-built chains become whole lanes, a block of the persistent grid streams down a lane segment,
    the upstream halo of every tile is what the same block has just written (hot in its XCD's L2) and two tile-local J^2
    applications carry information across the whole tile.  Against the Hilbert curve (compact tiles, chains of ~6 tiles whose
    concurrently active tiles are scattered over the XCD's region) -- measured in profiles/r03_c_chained_passes.txt.
    A field without a preferred axis keeps the Hilbert curve."""
    import os
    if os.environ.get('CWR_LANE_LEN'):                       # (A/B knob: tile length along the flow in cells; lanes are tile_rows / it wide)
        tile_len = max(1, int(os.environ['CWR_LANE_LEN']))
    (ax, ay), ratio = flow_axis(mesh, n_real)
    x = np.asarray(mesh['face_x'], dtype=np.float64)[:n_real]
    y = np.asarray(mesh['face_y'], dtype=np.float64)[:n_real]
    if ratio < min_ratio or n_real < 4 * tile_rows:
        return hilbert_order(mesh['face_x'], mesh['face_y'], n_real)
    s_along = x * ax + y * ay
    q_across = -x * ay + y * ax
    f1 = np.asarray(mesh['edges_face1'], dtype=np.int64)
    f2 = np.asarray(mesh['edges_face2'], dtype=np.int64)
    real = f2 < n_real
    h = float(np.median(np.hypot(x[f1[real]] - x[f2[real]], y[f1[real]] - y[f2[real]]))) if real.any() else 1.0
    width = max(1, tile_rows // tile_len) * max(h, 1e-300)
    lane = np.floor((q_across - q_across.min()) / width).astype(np.int64)
    # snake: odd lanes run against the axis, so that the end of a lane and the start of the next are neighbours (the tile that
    # straddles two lanes stays compact); the engine's chains follow the flow whatever the numbering direction
    key = np.where(lane & 1, -s_along, s_along)
    return np.lexsort((key, lane)).astype(np.int64)
