"""CPU model: a coarse level whose transfer operators FOLLOW THE FLOW -- the one accelerator profiles/r04_h did not
try.  r04_h's coarse level used piecewise-constant aggregates (P^T A P) and failed because the error a chained pass leaves is
transported error, which a piecewise-constant correction of an upwind operator cannot represent.  Here, in the style of approximate
ideal restriction (AIR) for non-symmetric M-matrices:

  C-points   the cells of every `agg`-th cross-section along the flow (the downstream end of a lane segment `agg` cells long),
  F-points   all other cells,
  R          = [ -A_cf Z   I ]   with  Z ~ A_ff^-1  taken from m terms of its Neumann series (Z exact = ideal restriction: then
               R A P is the Schur complement and the two-level cycle with exact F-relaxation is a direct solve),
  P          injection (coarse values at the C-points, nothing at the F-points), or the "upwind characteristic" -Z A_fc,
  coarse op  A_c = R A P, solved EXACTLY here (sparse LU): the best any coarse solver could do,
  smoother   the engine's chained in-place pass as modelled in chain_gs_probe.py / coarse_probe.py (lane-major numbering, 64-row tiles,
             lists of TPB tiles walked along the flow, L tile-local J^2 applications).

A cycle = one pass + one coarse correction (+ optionally one F-relaxation with Z).  Printed: cycles to a scaled residual of 1e-12,
beside the passes the smoother needs alone, and the size / fill of the coarse operator.
VERDICT's bar for building anything: >= 2 x fewer passes at CFL 25 (dt = 400) and no loss at CFL 2.5 (dt = 40).
usage: air_probe.py [nx] [dt ...]"""
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
dts = [float(v) for v in sys.argv[2:]] or [40.0, 400.0, 1000.0]
TR, TPB = 64, 15


def setup(dt):
    mesh = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=dt, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
    n = mesh['nreal'] + 1
    mesh = renumber_mesh(mesh, lane_order(mesh, n, tile_rows=TR))
    orc.derive_coefficients(mesh)
    lhs = orc.LHS(mesh); lhs.update_values(mesh, 0)
    A = lhs.csr().tocsr()[:n, :n]
    return mesh, n, A


def smoother(mesh, n, A, L):
    D = A.diagonal()
    J = (sp.identity(n, format='csr') - sp.diags(1.0 / D) @ A).tocsr()
    J.eliminate_zeros()
    J2 = (J @ J).tocsr()
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

    def one_pass(x, c2):
        for m in rounds:
            g = c2[m] + Jout[m] @ x
            y = x.copy()
            for _ in range(L):
                y[m] = g + (Jin[m] @ y)
            x = y
        return x
    return J, one_pass


def air_level(mesh, n, A, agg, m_terms, interp):
    """C/F splitting by cross-sections along the flow, R = [-A_cf Z, I], P, A_c = R A P (LU)."""
    (ax, ay), _ = flow_axis(mesh, n)
    s = np.asarray(mesh['face_x'])[:n] * ax + np.asarray(mesh['face_y'])[:n] * ay
    a0 = np.asarray(mesh['advection_coeff'][0], dtype=float)
    f1, f2 = np.asarray(mesh['edges_face1']), np.asarray(mesh['edges_face2'])
    real = f2 < n
    sgn = np.sign(np.sum(a0[real] * (s[f2[real]] - s[f1[real]]))) or 1.0
    s = sgn * s                                              # increasing downstream
    dx = (s.max() - s.min()) / (nx - 1 + 1e-9)
    col = np.floor((s - s.min()) / dx + 0.5).astype(int)
    isC = (col % agg) == (agg - 1)
    C, F = np.nonzero(isC)[0], np.nonzero(~isC)[0]
    Aff, Afc, Acf, Acc = A[F][:, F].tocsr(), A[F][:, C].tocsr(), A[C][:, F].tocsr(), A[C][:, C].tocsr()
    Dff = Aff.diagonal()
    if m_terms is None:                                      # ideal: Z = A_ff^-1 exactly (dense-ish: only for the model)
        lu_ff = splu(Aff.tocsc())
        Z_apply = lambda v: lu_ff.solve(v)
        ZT_apply = lambda v: lu_ff.solve(v, trans='T')
        Zmat = None
    else:
        Jff = (sp.identity(len(F), format='csr') - sp.diags(1.0 / Dff) @ Aff).tocsr()
        Zmat = sp.diags(1.0 / Dff).tocsr()
        term = Zmat.copy()
        for _ in range(m_terms - 1):
            term = (Jff @ term).tocsr()
            Zmat = (Zmat + term).tocsr()
        Z_apply = lambda v: Zmat @ v
    # R r = r_C - A_cf Z r_F ;  P e_c = [W e_c ; e_c] with W = 0 (injection) or -Z A_fc
    if Zmat is not None:
        RF = (-(Acf @ Zmat)).tocsr()                         # (nC x nF)
        W = (-(Zmat @ Afc)).tocsr() if interp == 'upwind' else None
        Ac = Acc + RF @ Afc
        if W is not None:
            Ac = Ac + (Acf + RF @ Aff) @ W
        fill = Ac.nnz / max(1, len(C))
        lu_c = splu(sp.csc_matrix(Ac))
        restrict = lambda r: r[C] + RF @ r[F]
    else:
        W = None
        # exact Schur complement, formed column by column would be dense: apply through solves instead (model only: LU of the whole A)
        lu_all = splu(A.tocsc())
        fill = float('nan')
        lu_c = None
        restrict = lambda r: r[C] - Acf @ Z_apply(r[F])
    return dict(C=C, F=F, restrict=restrict, lu_c=lu_c, W=W, Z=Z_apply, Afc=Afc, Aff=Aff, fill=fill,
                lu_all=locals().get('lu_all'), Acc=Acc, Acf=Acf)


def cycles_needed(mesh, n, A, L, lvl=None, post_f=False, maxp=300):
    J, one_pass = smoother(mesh, n, A, L)
    D = A.diagonal()
    rng = np.random.default_rng(0)
    xs = rng.uniform(1, 100, n)
    b = A @ xs
    bh = b / D
    c2 = bh + J @ bh
    x = xs * (1 + 0.3 * rng.standard_normal(n))
    nb = np.linalg.norm(bh)
    for p in range(1, maxp + 1):
        x = one_pass(x, c2)
        if lvl is not None:
            r = b - A @ x
            C, F = lvl['C'], lvl['F']
            rc = lvl['restrict'](r)
            if lvl['lu_c'] is not None:
                ec = lvl['lu_c'].solve(rc)
            else:                                            # ideal R: the Schur complement solve through the LU of A
                rhs = np.zeros(n); rhs[C] = rc
                ec = lvl['lu_all'].solve(rhs)[C]
            x[C] += ec
            if lvl['W'] is not None:
                x[F] += lvl['W'] @ ec
            if post_f:                                       # one F-relaxation with Z: x_F += Z (b - A x)_F
                r = b - A @ x
                x[F] += lvl['Z'](r[F])
        res = np.linalg.norm(bh - (x - J @ x)) / nb
        if not np.isfinite(res) or res > 1e30:
            return -p
        if res < 1e-12:
            return p
    return maxp


for dt in dts:
    mesh, n, A = setup(dt)
    D = A.diagonal()
    rho = float(np.max((abs(A).sum(axis=1).A1 - np.abs(D)) / D))
    L = 2 if rho < 0.9 else (3 if rho < 0.98 else 4)
    base = cycles_needed(mesh, n, A, L)
    print(f'n={n} dt={dt:g} ||J||_inf={rho:.4f} x{L}: chained passes alone {base}', flush=True)
    for agg in (4, 8):
        for m_terms, interp, post in ((None, 'inject', True), (agg + 2, 'inject', True), (agg + 2, 'upwind', False), (agg // 2 + 1, 'inject', True),
                                      (2, 'inject', True), (agg + 2, 'inject', False)):
            lvl = air_level(mesh, n, A, agg, m_terms, interp)
            pc = cycles_needed(mesh, n, A, L, lvl, post_f=post)
            name = 'ideal R' if m_terms is None else f'R from {m_terms} Neumann terms'
            # work in Jacobi-sweep equivalents over the fine mesh: a pass = 2 L, the residual 1, applying R = m sweeps over the F rows, the
            # F-relaxation m more; the coarse solve counted as FREE (the model solves it exactly) -- against 2 L per pass of the smoother alone
            mm = 0 if m_terms is None else m_terms
            work = abs(pc) * (2 * L + 1 + mm + (mm if post else 0))
            print(f'   every {agg}th cross-section coarse ({len(lvl["C"])} of {n}), {name}, P = {interp}{", + F-relaxation" if post else ""}: '
                  f'{pc if pc > 0 else "DIVERGED after " + str(-pc)} cycles; coarse operator {lvl["fill"]:.1f} entries per row; '
                  f'>= {work} sweep equivalents with a free coarse solve (smoother alone: {base * 2 * L})', flush=True)
