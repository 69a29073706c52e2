"""CPU model: does a coarse-level correction give the solver a CFL-independent component?

Smoother = the engine's chained in-place pass as modelled in chain_gs_probe.py (lane-major numbering, 64-row tiles, lists of
TPB tiles walked along the flow, L tile-local J^2 applications, tiles of one round see each other's old values).
Coarse level = AGGREGATES of cells with piecewise-constant prolongation P and restriction P^T on the UNSCALED system: summing the
finite-volume balances of an aggregate's cells gives the upwind balance of the aggregate (A_c = P^T A P is the same scheme on
the coarse mesh, an M-matrix, column sums V/dt), so the correction is mass-conservative by construction.  Aggregates tried:
the tiles themselves (4 cells long x 16 wide in lane-major order), blocks of `agg` consecutive cells ALONG a lane (1 wide), and
whole cross-sections.  The coarse system is solved exactly (spsolve): the best any coarse solver could do.

Printed: passes to a scaled residual of 1e-12 without / with the correction every `every` passes.
usage: coarse_probe.py [nx] [dt ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import splu

import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import lane_order, renumber_mesh
from clearwater_riverine_amd import schedule as sch
from oracle import cwr_oracle as orc

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 250
dts = [float(v) for v in sys.argv[2:]] or [40.0, 400.0, 1000.0, 3600.0]
TR, TPB = 64, 15


def setup(dt):
    mesh = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=dt, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
    n = mesh['nreal'] + 1
    mesh = renumber_mesh(mesh, lane_order(mesh, n, tile_rows=TR))
    orc.derive_coefficients(mesh)
    lhs = orc.LHS(mesh); lhs.update_values(mesh, 0)
    A = lhs.csr().tocsr()[:n, :n]
    return mesh, n, A


def passes_needed(mesh, n, A, L, agg_of=None, every=1, maxp=400):
    D = A.diagonal()
    J = sp.identity(n, format='csr') - sp.diags(1.0 / D) @ A
    J.eliminate_zeros()
    J2 = (J @ J).tocsr()
    rng = np.random.default_rng(0)
    xs = rng.uniform(1, 100, n)
    bh = xs - J @ xs
    c2 = bh + J @ bh
    x = xs * (1 + 0.3 * rng.standard_normal(n))
    nb = np.linalg.norm(bh)
    tile = np.arange(n) // TR
    ntiles = int(tile.max()) + 1
    NB = max(8, (ntiles // TPB) // 8 * 8)
    coo = J2.tocoo()
    inside = tile[coo.row] == tile[coo.col]
    Jin = sp.csr_matrix((coo.data[inside], (coo.row[inside], coo.col[inside])), shape=(n, n)).tocsr()
    Jout = sp.csr_matrix((coo.data[~inside], (coo.row[~inside], coo.col[~inside])), shape=(n, n)).tocsr()
    rows_of = [np.arange(t * TR, min((t + 1) * TR, n)) for t in range(ntiles)]
    sched = sch.chain_schedule(mesh['edges_face1'], mesh['edges_face2'], mesh['advection_coeff'][0], n, TR, ntiles, NB, streams_per_block=1)
    rounds = [np.concatenate([rows_of[t] for t in row if t >= 0]) for row in sched if (row >= 0).any()]
    coarse = None
    if agg_of is not None:
        nc = int(agg_of.max()) + 1
        P = sp.csr_matrix((np.ones(n), (np.arange(n), agg_of)), shape=(n, nc))
        Ac = (P.T @ A @ P).tocsc()
        coarse = (P, splu(Ac), nc)
    for p in range(1, maxp + 1):
        for m in rounds:
            g = c2[m] + Jout[m] @ x
            y = x.copy()
            for _ in range(L):
                y[m] = g + (Jin[m] @ y)
            x = y
        if coarse is not None and p % every == 0:
            P, lu, _ = coarse
            r_u = D * (bh - (x - J @ x))                     # unscaled residual b - A x
            x = x + P @ lu.solve(P.T @ r_u)
        r = np.linalg.norm(bh - (x - J @ x)) / nb
        if r < 1e-12:
            return p
    return maxp


for dt in dts:
    mesh, n, A = setup(dt)
    D = A.diagonal()
    rho = float(np.max((abs(A).sum(axis=1).A1 - np.abs(D)) / D))
    L = 2 if rho < 0.9 else (4 if rho < 0.98 else (6 if rho < 0.993 else 8))
    x, y = np.asarray(mesh['face_x'])[:n], np.asarray(mesh['face_y'])[:n]
    tile = np.arange(n) // TR
    # aggregates along a lane: consecutive cells of the lane-major numbering share a lane; a lane is TR / 4 = 16 cells wide, so
    # `agg` consecutive tiles are one strip 4 agg cells long
    aggs = {'tiles (4 x 16 cells)': tile, '2 tiles (8 x 16)': tile // 2, '4 tiles (16 x 16)': tile // 4}
    base = passes_needed(mesh, n, A, L)
    line = f'n={n} dt={dt:g} ||J||_inf={rho:.4f} x{L}: chained passes alone {base}'
    for name, agg in aggs.items():
        for every in (1, 2):
            pc = passes_needed(mesh, n, A, L, agg, every)
            line += f'; + {name} every {every}: {pc} ({int(agg.max()) + 1} aggregates)'
    print(line, flush=True)
