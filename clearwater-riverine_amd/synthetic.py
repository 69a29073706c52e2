"""Synthetic HEC-RAS-2D-like meshes and flow fields (host side, numpy only).

The Ohio River / Sumwere HDF files of the reference are missing blobs, and the
1 M / 4 M-cell benchmark meshes never existed, so the workloads of BASELINE.json are
generated here with the array surface the reference's HDF reader yields
(/root/reference/src/clearwater_riverine/io/hdf.py:246-310):

    edges_face1, edges_face2  (E,) int32   face1 always a real cell, ghost cells only as face2,
                                           one ghost cell per perimeter face, ids > nreal
    face_x, face_y            (ncell,) f64
    face_flow, edge_velocity  (T, E) f32   signed: > 0 flows face1 -> face2
    volume                    (T, ncell) f32
    time_seconds              (T,) f64

Construction: an nx x ny quad band with jittered nodes; optionally
`n_merge` horizontally adjacent cell pairs are fused into 6-sided cells (mixed cell
degrees, duplicate faces between the same two cells); cells and faces are renumbered
row-major with a random shuffle inside windows of `shuffle_window` ids and random face
orientation (HEC-RAS numbering is not monotone).  The flow is a node stream function
(exactly divergence-free per cell: uniform through-flow + wall-bounded eddies, with a
sinusoidal unsteadiness) plus a small potential component whose divergence is integrated
into the cell volumes, so the float64 field satisfies discrete continuity
V[t+1] - V[t] = -dt * sum(signed outflow) before it is rounded to float32.
Walls carry exactly zero flow and zero velocity (so, as in the reference, no diffusion).
"""
from __future__ import annotations

import numpy as np


def make_mesh(nx: int, ny: int, n_steps: int, *, seed: int = 0, n_merge: int = 0,
              shuffle_window: int = 32, dx: float = 10.0, dy: float = 10.0, depth: float = 2.0,
              dt: float = 10.0, velocity: float = 0.5, unsteady: float = 0.1, period_steps: int = 24,
              eddy: float = 0.3, breathing: float = 0.02, jitter: float = 0.2,
              diffusion_coefficient: float = 0.1, n_dry: int = 0, steady: bool = False, n_merge4: int = 0) -> dict:
    """Return a mesh dict (reference variable names) with T = n_steps + 1 time levels.
    steady=True freezes the field in time (then the reference scheme is exactly conservative).
    n_merge pairs of quads become 6-sided cells; n_merge4 blocks of 2 x 2 quads become 8-sided cells (HEC-RAS allows up to
    8 faces per cell).  Faces between the same two cells stay separate faces, as in HEC-RAS output."""
    rng = np.random.default_rng(seed)
    if steady:
        unsteady = 0.0
        breathing = 0.0
    T = n_steps + 1

    # ---- base grid: cell (i, j) -> base id j*nx + i ; optional pair merges ------------
    nb = nx * ny
    base_to_cell = np.arange(nb, dtype=np.int64)
    in_block = np.zeros(nb, dtype=bool)
    if n_merge4 > 0:
        # 2 x 2 blocks at even (i, j) never overlap; drawn from a generator of their own so that meshes without them
        # keep the random stream (and every fixture) they had before the option existed
        rng4 = np.random.default_rng(seed + 7919)
        i4, j4 = np.meshgrid(np.arange(0, nx - 1, 2), np.arange(0, ny - 1, 2), indexing='xy')
        cand4 = (j4 * nx + i4).ravel()
        if n_merge4 > len(cand4):
            raise ValueError('n_merge4 too large for this grid')
        pick4 = rng4.choice(cand4, size=n_merge4, replace=False)
        for off in (1, nx, nx + 1):
            base_to_cell[pick4 + off] = pick4
        for off in (0, 1, nx, nx + 1):
            in_block[pick4 + off] = True
    if n_merge > 0:
        # candidate pairs ((2m, j), (2m+1, j)) never overlap
        ii, jj = np.meshgrid(np.arange(0, nx - 1, 2), np.arange(ny), indexing='xy')
        cand = (jj * nx + ii).ravel()
        if n_merge4 > 0:
            cand = cand[~in_block[cand]]
        if n_merge > len(cand):
            raise ValueError('n_merge too large for this grid')
        pick = rng.choice(cand, size=n_merge, replace=False)
        base_to_cell[pick + 1] = pick                      # right cell joins the left one
    # compact + locally shuffled numbering of the surviving cells
    reps = np.unique(base_to_cell)
    n = len(reps)
    order = np.arange(n)
    if shuffle_window > 1:
        keys = (order // shuffle_window).astype(np.float64) + rng.random(n)
        order = np.argsort(keys, kind='stable')            # position -> old compact id
    newid = np.empty(n, dtype=np.int64)
    newid[order] = np.arange(n)
    compact = np.full(nb, -1, dtype=np.int64)
    compact[reps] = newid
    cell_of_base = compact[base_to_cell]                   # (nb,) final real-cell id

    # ---- nodes (jittered) and the two families of faces --------------------------------
    xn = np.arange(nx + 1, dtype=np.float64)[None, :] * dx + np.zeros((ny + 1, 1))
    yn = np.arange(ny + 1, dtype=np.float64)[:, None] * dy + np.zeros((1, nx + 1))
    if jitter > 0:
        jx = (rng.random((ny + 1, nx + 1)) - 0.5) * 2 * jitter * dx
        jy = (rng.random((ny + 1, nx + 1)) - 0.5) * 2 * jitter * dy
        jx[:, 0] = jx[:, -1] = 0.0
        jy[0, :] = jy[-1, :] = 0.0
        jx[0, :] = jx[-1, :] = 0.0
        jy[:, 0] = jy[:, -1] = 0.0
        xn = xn + jx
        yn = yn + jy

    def base_id(i, j):
        return j * nx + i

    # vertical faces: between base cells (i-1, j) and (i, j), i = 0..nx ; node (i, j)-(i, j+1)
    iv, jv = np.meshgrid(np.arange(nx + 1), np.arange(ny), indexing='xy')
    iv = iv.ravel(); jv = jv.ravel()
    v_left = np.where(iv > 0, base_id(np.maximum(iv - 1, 0), jv), -1)
    v_right = np.where(iv < nx, base_id(np.minimum(iv, nx - 1), jv), -1)
    v_len = np.hypot(xn[jv + 1, iv] - xn[jv, iv], yn[jv + 1, iv] - yn[jv, iv])
    # horizontal faces: between base cells (i, j-1) and (i, j), j = 0..ny ; node (i, j)-(i+1, j)
    ih, jh = np.meshgrid(np.arange(nx), np.arange(ny + 1), indexing='xy')
    ih = ih.ravel(); jh = jh.ravel()
    h_lo = np.where(jh > 0, base_id(ih, np.maximum(jh - 1, 0)), -1)
    h_hi = np.where(jh < ny, base_id(ih, np.minimum(jh, ny - 1)), -1)
    h_len = np.hypot(xn[jh, ih + 1] - xn[jh, ih], yn[jh, ih + 1] - yn[jh, ih])

    # positive direction: vertical faces left -> right, horizontal faces low -> high
    side_a = np.concatenate([v_left, h_lo])                 # upstream side of the + direction
    side_b = np.concatenate([v_right, h_hi])
    flen = np.concatenate([v_len, h_len])
    is_vert = np.concatenate([np.ones(len(v_left), bool), np.zeros(len(h_lo), bool)])
    node_i = np.concatenate([iv, ih]); node_j = np.concatenate([jv, jh])

    ca = np.where(side_a >= 0, cell_of_base[np.maximum(side_a, 0)], -1)
    cb = np.where(side_b >= 0, cell_of_base[np.maximum(side_b, 0)], -1)
    keep = ca != cb                                         # drop faces swallowed by a merge
    ca, cb, flen, is_vert = ca[keep], cb[keep], flen[keep], is_vert[keep]
    node_i, node_j = node_i[keep], node_j[keep]
    E = len(ca)

    # ---- face numbering: spatial order with a windowed shuffle --------------------------
    pos_key = np.where(ca >= 0, ca, cb).astype(np.float64)  # near the owning cells' ids
    eorder = np.argsort(pos_key + rng.random(E) * max(shuffle_window, 1), kind='stable')
    ca, cb, flen, is_vert = ca[eorder], cb[eorder], flen[eorder], is_vert[eorder]
    node_i, node_j = node_i[eorder], node_j[eorder]

    # ---- orientation: face1 real; perimeter faces get their own ghost cell as face2 ------
    boundary = (ca < 0) | (cb < 0)
    flip = np.where(boundary, ca < 0, rng.random(E) < 0.5)  # True: face1 = b side, + dir is face2->face1
    face1 = np.where(flip, cb, ca)
    face2 = np.where(flip, ca, cb)
    n_ghost = int(boundary.sum())
    face2 = face2.copy()
    face2[boundary] = n + np.arange(n_ghost)
    sign = np.where(flip, -1.0, 1.0)                        # flow(face1->face2) = sign * flow(+dir)
    ncell = n + n_ghost

    # ---- cell centres and volumes --------------------------------------------------------
    bi, bj = np.meshgrid(np.arange(nx), np.arange(ny), indexing='xy')
    bcx = 0.25 * (xn[:-1, :-1] + xn[:-1, 1:] + xn[1:, :-1] + xn[1:, 1:]).ravel()
    bcy = 0.25 * (yn[:-1, :-1] + yn[:-1, 1:] + yn[1:, :-1] + yn[1:, 1:]).ravel()
    cnt = np.bincount(cell_of_base, minlength=n).astype(np.float64)
    face_x = np.empty(ncell); face_y = np.empty(ncell)
    face_x[:n] = np.bincount(cell_of_base, weights=bcx, minlength=n) / cnt
    face_y[:n] = np.bincount(cell_of_base, weights=bcy, minlength=n) / cnt
    # ghost centre: mirror of the real centre through the face midpoint (outside the band)
    gi, gj = node_i[boundary], node_j[boundary]
    gv = is_vert[boundary]
    mx = np.where(gv, xn[gj, gi], 0.5 * (xn[gj, gi] + xn[gj, np.minimum(gi + 1, nx)]))
    my = np.where(gv, 0.5 * (yn[gj, gi] + yn[np.minimum(gj + 1, ny), gi]), yn[gj, gi])
    rp = face1[boundary]
    face_x[n:] = mx + 0.5 * (mx - face_x[rp])
    face_y[n:] = my + 0.5 * (my - face_y[rp])
    area0 = cnt * dx * dy * (1.0 + 0.1 * (rng.random(n) - 0.5))
    vol0 = area0 * depth

    # ---- flow field ------------------------------------------------------------------------
    # stream function at nodes (ny+1, nx+1): through-flow Q per unit row + eddies vanishing on walls
    q_row = velocity * depth * dy                           # flow through one vertical face
    jn = np.arange(ny + 1, dtype=np.float64)[:, None]
    inn = np.arange(nx + 1, dtype=np.float64)[None, :]
    psi_uniform = q_row * jn + np.zeros((1, nx + 1))
    mx_modes = max(1, nx // 12)
    psi_eddy = eddy * q_row * min(ny, 8) / np.pi * \
        np.sin(np.pi * inn / nx * mx_modes) * np.sin(np.pi * jn / ny)
    # + direction flux: vertical face psi(i, j+1) - psi(i, j); horizontal face -(psi(i+1, j) - psi(i, j))
    def plus_flux(psi):
        fv = psi[node_j + 1 * is_vert, node_i] - psi[node_j, node_i]
        fh = -(psi[node_j, np.minimum(node_i + 1, nx)] - psi[node_j, node_i])
        return np.where(is_vert, fv, fh)
    f_uniform = plus_flux(psi_uniform)
    f_eddy = plus_flux(psi_eddy)
    # potential ("breathing") component on internal faces only
    internal = ~boundary
    kappa = flen * depth / np.where(is_vert, dx, dy)
    phase = 2 * np.pi * (face_x[:n] / max(nx * dx, 1.0))
    p1 = np.where(internal, face1, 0); p2 = np.where(internal, np.minimum(face2, n - 1), 0)

    steps = np.arange(T, dtype=np.float64)
    amp_t = 1.0 + unsteady * np.sin(2 * np.pi * steps / period_steps)
    flow = np.empty((T, E), dtype=np.float64)
    vol = np.empty((T, n), dtype=np.float64)
    vol[0] = vol0
    breath_scale = breathing * vol0.mean() / dt / 4.0
    for t in range(T):
        phi = breath_scale * np.sin(2 * np.pi * steps[t] / (period_steps * 0.75) + phase)
        pot = np.where(internal, 0.25 * (phi[p1] - phi[p2]) * kappa / kappa.mean(), 0.0)
        eddy_t = 1.0 if steady else np.cos(2 * np.pi * steps[t] / (2 * period_steps))
        f12 = sign * (amp_t[t] * f_uniform + eddy_t * f_eddy) + pot
        flow[t] = f12
        if t + 1 < T:
            div = np.bincount(face1, weights=f12, minlength=n) - \
                np.bincount(np.where(internal, face2, 0), weights=np.where(internal, f12, 0.0), minlength=n)
            vol[t + 1] = vol[t] - dt * div
    if vol.min() <= 0:
        raise ValueError('synthetic volumes went non-positive; lower dt/velocity/breathing')
    flow[np.abs(flow) < 1e-12 * q_row] = 0.0                # walls: exactly zero
    vel = flow / (flen * depth)[None, :]

    volume = np.zeros((T, ncell), dtype=np.float32)
    volume[:, :n] = vol
    flow32 = flow.astype(np.float32)
    vel32 = vel.astype(np.float32)
    vel32[flow32 == 0] = 0.0

    if n_dry > 0:                                           # permanently dry cells: V = 0, no flow on their faces
        dry = rng.choice(n, size=n_dry, replace=False)
        volume[:, dry] = 0.0
        isdry = np.zeros(ncell, bool); isdry[dry] = True
        touch = isdry[face1] | isdry[np.minimum(face2, ncell - 1)]
        flow32[:, touch] = 0.0
        vel32[:, touch] = 0.0

    inlet = boundary & is_vert & (node_i == 0)
    outlet = boundary & is_vert & (node_i == nx)
    return {
        'edges_face1': face1.astype(np.int32), 'edges_face2': face2.astype(np.int32),
        'nreal': int(n - 1),
        'face_x': face_x, 'face_y': face_y,
        'face_flow': flow32, 'edge_velocity': vel32, 'volume': volume,
        'time_seconds': steps * dt,
        'diffusion_coefficient': float(diffusion_coefficient),
        'inlet_ghost_cells': face2[inlet].astype(np.int64),
        'outlet_ghost_cells': face2[outlet].astype(np.int64),
        'wall_ghost_cells': face2[boundary & ~inlet & ~outlet].astype(np.int64),
        'grid': (nx, ny),
    }


def bend_channel(mesh: dict, theta0: float = 1.0, wavelength: float | None = None) -> dict:
    """The same mesh laid along a meander: the x axis of make_mesh becomes the arc length of a sine-generated centre line (direction
    angle theta0 * sin(2 pi s / wavelength), the textbook river meander), y the offset along its normal.  Topology, flows and
    volumes are untouched -- only the cell-centre coordinates move, and with them what depends on them: the internal numbering
    (ordering.py) and the face-to-face distances of the diffusion term.  theta0 in radians (1.0: the channel swings +-57 degrees);
    the wavelength defaults to the channel length and must keep the radius of curvature above half the channel width."""
    x = np.asarray(mesh['face_x'], dtype=np.float64)
    y = np.asarray(mesh['face_y'], dtype=np.float64)
    L = float(x.max() - x.min())
    lam = float(wavelength) if wavelength else L
    half = 0.5 * float(y.max() - y.min())
    if theta0 * 2 * np.pi / lam * half >= 0.9:
        raise ValueError('bend_channel: the inner bank would fold (radius of curvature below half the channel width)')
    sg = np.linspace(x.min(), x.max(), 20001)
    th = theta0 * np.sin(2 * np.pi * (sg - sg[0]) / lam)
    ds = sg[1] - sg[0]
    cx = np.concatenate([[0.0], np.cumsum(0.5 * (np.cos(th[1:]) + np.cos(th[:-1])) * ds)])
    cy = np.concatenate([[0.0], np.cumsum(0.5 * (np.sin(th[1:]) + np.sin(th[:-1])) * ds)])
    # (ghost centres lie just outside [x.min, x.max]: clamp the arc length, extend along the end tangents)
    sc = np.clip(x, sg[0], sg[-1])
    t_ = np.interp(sc, sg, th)
    px = np.interp(sc, sg, cx) + (x - sc) * np.cos(t_)
    py = np.interp(sc, sg, cy) + (x - sc) * np.sin(t_)
    off = y - 0.5 * (y.max() + y.min())
    m = dict(mesh)
    m['face_x'] = px - off * np.sin(t_)
    m['face_y'] = py + off * np.cos(t_)
    m.pop('face_to_face_dist', None)
    return m


BENCH_SEED = 4
BENCH_GRID = (1026, 1026)
BENCH_MERGES = 52676            # 5.0 % of the 1 052 676 base quads fused into 6-sided cells: exactly 10^6 real cells


def bench_mesh(n_steps: int, *, dt: float = 40.0, diffusion_coefficient: float = 0.5, scale: int = 1) -> dict:
    """BASELINE config 4 / the bench workload: the unstructured 1 M-cell floodplain mesh --
    jittered quads with 5 % of the cells merged into 6-sided cells (duplicate faces between the same two cells
    included), locally shuffled cell and face numbering, seed 4, CFL ~ 2.5 at dt = 40 s, D = 0.5.
    scale = 2 gives config 5's 4 M-cell mesh (2052 x 2052 base quads, seed 5)."""
    nx, ny = BENCH_GRID[0] * scale, BENCH_GRID[1] * scale
    return make_mesh(nx, ny, n_steps, seed=BENCH_SEED if scale == 1 else BENCH_SEED + scale - 1,
                     n_merge=BENCH_MERGES * scale * scale, dt=dt, diffusion_coefficient=diffusion_coefficient)


def boundary_input_array(mesh: dict, n_const: int, *, ic: float | np.ndarray = 1.0,
                         inlet_base: float = 60.0, inlet_amp: float = 40.0,
                         inlet_period_s: float = 600.0, outlet_value: float = 5.0) -> np.ndarray:
    """(T, ncell, K) input array in the reference's convention
    (/root/reference/src/clearwater_riverine/constituents.py:78-164): row 0 carries the
    initial condition of the real cells, ghost-cell columns carry boundary values, zero
    means "no boundary value".  Constituent k is scaled by (k + 1)."""
    T = len(mesh['time_seconds'])
    ncell = len(mesh['face_x'])
    n = mesh['nreal'] + 1
    arr = np.zeros((T, ncell, n_const))
    scale = (np.arange(n_const) + 1.0)[None, :]
    ic_arr = np.broadcast_to(np.asarray(ic, dtype=np.float64).reshape(-1, 1) if np.ndim(ic) else
                             np.full((n, 1), float(ic)), (n, 1))
    arr[0, :n, :] = ic_arr * scale
    tsec = np.asarray(mesh['time_seconds'])
    series = inlet_base + inlet_amp * np.sin(2 * np.pi * tsec / inlet_period_s)
    arr[:, mesh['inlet_ghost_cells'], :] = series[:, None, None] * scale[None]
    if outlet_value:
        out = mesh['outlet_ghost_cells']
        arr[:, out[::2], :] = outlet_value * scale[None]   # every other outlet ghost: the rest have "no BC"
    return arr


class DistinctInputs:
    """The input array of distinct_input_array WITHOUT materialising it: (T, ncell, K) float64 is 3.3 GB at 1 M cells x 16 x 26
    levels, and a rank of a partitioned run needs its own cells' initial rows and the boundary series of the ghost cells it holds.
    Every constituent has its own initial field and its own boundary series, so that the K batched systems have different
    right-hand sides, fronts and convergence histories.  Four families by k % 4:
      0  smooth field  A (1 + 0.5 sin(2 pi p x / Lx + phi) cos(2 pi q y / Ly)), sinusoidal inlet (period, phase by k)
      1  uniform field, inlet = base + a Gaussian pulse in time (a spill / E. coli spike)
      2  linear gradient along x, constant inlet, no outlet value
      3  plume: EXACTLY zero outside a disc (so the implicit solution has fronts decaying to 1e-300), small
         constant inlet value -- the case an element-wise parity check exists for
    Every value is >= 0; zero on a ghost cell keeps its reference meaning "no boundary value" (constituents.py:78-164 convention).
    distributed.PartitionedTransport takes an instance in place of the dense array (initial_rows / ghost_columns / real_input_entries)."""

    def __init__(self, mesh: dict, n_const: int, *, seed: int = 0):
        rng = np.random.default_rng(seed)
        self.T = len(mesh['time_seconds'])
        self.ncell = len(mesh['face_x'])
        self.K = int(n_const)
        self.n = mesh['nreal'] + 1
        self.shape = (self.T, self.ncell, self.K)
        n = self.n
        x = np.asarray(mesh['face_x'][:n], dtype=np.float64)
        y = np.asarray(mesh['face_y'][:n], dtype=np.float64)
        x0, x1, y0, y1 = x.min(), x.max(), y.min(), y.max()
        self.Lx, self.Ly = max(x1 - x0, 1.0), max(y1 - y0, 1.0)
        self._xs, self._ys = (x - x0) / self.Lx, (y - y0) / self.Ly
        tsec = np.asarray(mesh['time_seconds'], dtype=np.float64)
        span = max(tsec[-1] - tsec[0], 1.0)
        self.inlet = np.asarray(mesh['inlet_ghost_cells'], dtype=np.int64)
        self.outlet = np.asarray(mesh['outlet_ghost_cells'], dtype=np.int64)
        self._par = []                                   # per constituent: (family, amplitude, phase, extras)
        self._series = np.zeros((self.T, self.K))        # inlet series
        self._out_val = np.zeros(self.K)                 # value on the outlet ghosts the constituent uses ...
        self._out_sel = [None] * self.K                  # ... and which of them (slice of self.outlet)
        for k in range(self.K):
            fam = k % 4
            amp = 1.0 + 0.75 * k
            phi = 2 * np.pi * rng.random()
            extra = None
            if fam == 0:
                extra = (1 + (k // 4) % 3, 1 + (k // 8) % 2)
                self._series[:, k] = amp * (60.0 + 40.0 * np.sin(2 * np.pi * tsec / (600.0 * (1 + 0.25 * k)) + phi))
                self._out_val[k], self._out_sel[k] = 5.0 * amp, slice(0, None, 2)
            elif fam == 1:
                tc = tsec[0] + (0.3 + 0.4 * rng.random()) * span
                self._series[:, k] = amp * (2.0 + 80.0 * np.exp(-0.5 * ((tsec - tc) / (0.15 * span)) ** 2))
                self._out_val[k], self._out_sel[k] = 0.5 * amp, slice(1, None, 2)
            elif fam == 2:
                self._series[:, k] = 0.2 * amp
            else:
                extra = (0.2 + 0.6 * rng.random(), 0.2 + 0.6 * rng.random())
                self._series[:, k] = 1e-3 * amp
            self._par.append((fam, amp, phi, extra))

    def initial_rows(self, cells) -> np.ndarray:
        """(len(cells), K): row 0 of the input array on these REAL cells (the initial condition)."""
        cells = np.asarray(cells, dtype=np.int64)
        xs, ys = self._xs[cells], self._ys[cells]
        out = np.empty((len(cells), self.K))
        for k, (fam, amp, phi, extra) in enumerate(self._par):
            if fam == 0:
                p, q = extra
                out[:, k] = amp * (1.0 + 0.5 * np.sin(2 * np.pi * p * xs + phi) * np.cos(2 * np.pi * q * ys))
            elif fam == 1:
                out[:, k] = amp
            elif fam == 2:
                out[:, k] = amp * (0.2 + 2.0 * xs)
            else:
                cx, cy = extra
                r2 = ((xs - cx) * self.Lx) ** 2 + ((ys - cy) * self.Ly) ** 2
                rad = 0.08 * min(self.Lx, self.Ly)
                out[:, k] = np.where(r2 < rad * rad, 50.0 * amp * np.exp(-4.0 * r2 / (rad * rad)), 0.0)
        return out

    def ghost_columns(self, ghost_cells) -> np.ndarray:
        """(T, len(ghost_cells), K): the boundary series on these ghost cells (zero: no boundary value)."""
        g = np.asarray(ghost_cells, dtype=np.int64)
        out = np.zeros((self.T, len(g), self.K))
        is_in = np.isin(g, self.inlet)
        out[:, is_in, :] = self._series[:, None, :]
        for k in range(self.K):
            if self._out_sel[k] is not None:
                out[:, np.isin(g, self.outlet[self._out_sel[k]]), k] = self._out_val[k]
        return out

    def real_input_entries(self, cells):
        """Non-zero entries on real cells at levels >= 1 (none: the initial row is the only one that carries real-cell values)."""
        return np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros((0, self.K))

    def column(self, k: int) -> np.ndarray:
        """(T, ncell): constituent k of the dense array."""
        arr = np.zeros((self.T, self.ncell))
        one = self.initial_rows(np.arange(self.n))[:, k] if self.K == 1 else None
        arr[0, :self.n] = one if one is not None else self._initial_column(k)
        arr[:, self.n:] = self.ghost_columns(np.arange(self.n, self.ncell))[:, :, k]
        return arr

    def _initial_column(self, k: int) -> np.ndarray:
        sub = DistinctInputs.__new__(DistinctInputs)
        sub.__dict__.update(self.__dict__)
        sub.K, sub._par = 1, [self._par[k]]
        return sub.initial_rows(np.arange(self.n))[:, 0]

    def dense(self) -> np.ndarray:
        arr = np.zeros(self.shape)
        arr[0, :self.n, :] = self.initial_rows(np.arange(self.n))
        arr[:, self.n:, :] = self.ghost_columns(np.arange(self.n, self.ncell))
        return arr


def distinct_input_array(mesh: dict, n_const: int, *, seed: int = 0) -> np.ndarray:
    """(T, ncell, K) input array (constituents.py:78-164 convention, as boundary_input_array) in which every constituent has its
    own initial field and its own boundary series: DistinctInputs, materialised."""
    return DistinctInputs(mesh, n_const, seed=seed).dense()
