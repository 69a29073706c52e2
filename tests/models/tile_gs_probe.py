"""CPU model: what a stronger TILE-LOCAL solve would buy the chained passes.

Today a tile visit applies J^2 L times to the tile's own rows with its halo frozen (2 L Jacobi sweeps, every row at once).  Modelled:
  j2 x L        today
  gs x S        S Gauss-Seidel sweeps inside the tile, the rows taken in 4 STAGES along the flow (a 64-row tile is 4 cells long x 16
                wide in the lane-major numbering: a stage = one cross-section of 16 cells, Jacobi among themselves), halo frozen
  exact         the tile's rows solved exactly for the frozen halo (upper bound of any tile-local method)
Everything else as the engine runs it: lane-major numbering, 64-row tiles, lists of 15 tiles chained along the flow, tiles of one
round see each other's old values.  Printed: passes to a scaled residual of 1e-12.
usage: tile_gs_probe.py [nx] [dt ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import splu

import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import flow_axis, lane_order, renumber_mesh
from clearwater_riverine_amd import schedule as sch
from oracle import cwr_oracle as orc

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 160
dts = [float(v) for v in sys.argv[2:]] or [40.0, 400.0]
TR, TPB = 64, 15
for dt in dts:
    mesh = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=dt, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
    n = mesh['nreal'] + 1
    mesh = renumber_mesh(mesh, lane_order(mesh, n, tile_rows=TR))
    orc.derive_coefficients(mesh)
    lhs = orc.LHS(mesh); lhs.update_values(mesh, 0)
    A = lhs.csr().tocsr()[:n, :n]
    D = A.diagonal()
    J = (sp.identity(n, format='csr') - sp.diags(1.0 / D) @ A).tocsr()
    J.eliminate_zeros()
    J2 = (J @ J).tocsr()
    rho = float(np.max((abs(A).sum(axis=1).A1 - np.abs(D)) / D))
    L = 2 if rho < 0.9 else (4 if rho < 0.98 else (6 if rho < 0.993 else 8))
    rng = np.random.default_rng(0)
    xs = rng.uniform(1, 100, n)
    bh = xs - J @ xs
    c2 = bh + J @ bh
    x0 = xs * (1 + 0.3 * rng.standard_normal(n))
    nb = np.linalg.norm(bh)
    tile = np.arange(n) // TR
    ntiles = int(tile.max()) + 1
    NB = max(8, (ntiles // TPB) // 8 * 8)
    rows_of = [np.arange(t * TR, min((t + 1) * TR, n)) for t in range(ntiles)]
    sched = sch.chain_schedule(mesh['edges_face1'], mesh['edges_face2'], mesh['advection_coeff'][0], n, TR, ntiles, NB, streams_per_block=1)
    rounds = [[int(t) for t in row if t >= 0] for row in sched if (row >= 0).any()]
    # stages of every tile: its rows in 4 groups along the flow (sign of the net flow along the axis decides the direction)
    (ax, ay), _ = flow_axis(mesh, n)
    s_along = np.asarray(mesh['face_x'])[:n] * ax + np.asarray(mesh['face_y'])[:n] * ay
    a0 = np.asarray(mesh['advection_coeff'][0], dtype=float)
    f1, f2 = np.asarray(mesh['edges_face1']), np.asarray(mesh['edges_face2'])
    real = f2 < n
    sgn = np.sign(np.sum(a0[real] * (s_along[f2[real]] - s_along[f1[real]])))
    stages = []
    for t in range(ntiles):
        r = rows_of[t]
        o = r[np.argsort(sgn * s_along[r], kind='stable')]
        stages.append([o[i:i + 16] for i in range(0, len(o), 16)])
    Jc = J.tocsr(); J2c = J2.tocsr()
    lus = None

    def run(kind, S, maxp=300):
        global lus
        x = x0.copy()
        if kind == 'exact' and lus is None:
            Ahat = (sp.identity(n, format='csr') - J).tocsr()
            lus = [splu(Ahat[r][:, r].tocsc()) for r in rows_of]
        for p in range(1, maxp + 1):
            for rnd in rounds:
                xin = x.copy()                                   # tiles of one round see each other's old values
                for t in rnd:
                    r = rows_of[t]
                    y = xin.copy()
                    if kind == 'j2':
                        for _ in range(S):
                            y[r] = c2[r] + J2c[r] @ y
                    elif kind == 'gs':
                        for _ in range(S):
                            for st in stages[t]:
                                y[st] = bh[st] + Jc[st] @ y
                    else:
                        rhs = bh[r] + Jc[r] @ y - (Jc[r][:, r] @ y[r])
                        y[r] = lus[t].solve(rhs)
                    x[r] = y[r]
            if np.linalg.norm(bh - (x - J @ x)) / nb < 1e-12:
                return p
        return maxp

    line = f'n={n} dt={dt:g} ||J||_inf={rho:.4f}: passes  j2 x{L} (today) {run("j2", L)}'
    for S in (1, 2, 3, 4):
        line += f'; gs x{S} {run("gs", S)}'
    line += f'; exact tile solve {run("exact", 0)}'
    print(line, flush=True)
    lus = None
