"""CPU model: the deterministic list-walking pass (round 4: ping-pong between two vectors, a tile takes the rows of the earlier tiles of
its own list fresh and every other row from the pass's input) is a FIXED linear operator -- so it can precondition a Krylov method.
Modelled: stationary passes against BiCGSTAB and restarted GMRES with M^-1 r = p passes on A z = r from z = 0, on the lane-major
probe mesh.  Cost unit: one pass (a BiCGSTAB iteration = 2 p passes + 2 c2 sweeps + 2 products + vector work, printed separately).
usage: krylov_probe.py [nx] [dt ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import lane_order, renumber_mesh
from clearwater_riverine_amd import schedule as sch
from oracle import cwr_oracle as orc

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 160
dts = [float(v) for v in sys.argv[2:]] or [40.0, 400.0, 1000.0]
TR, TPB = 64, 15
for dt in dts:
    mesh = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=dt, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
    n = mesh['nreal'] + 1
    mesh = renumber_mesh(mesh, lane_order(mesh, n, tile_rows=TR))
    orc.derive_coefficients(mesh)
    lhs = orc.LHS(mesh); lhs.update_values(mesh, 0)
    A = lhs.csr().tocsr()[:n, :n]
    D = A.diagonal()
    Ah = (sp.diags(1.0 / D) @ A).tocsr()
    J = (sp.identity(n, format='csr') - Ah).tocsr()
    J.eliminate_zeros()
    J2 = (J @ J).tocsr()
    rho = float(np.max((abs(A).sum(axis=1).A1 - np.abs(D)) / D))
    L = 2 if rho < 0.9 else (4 if rho < 0.98 else (6 if rho < 0.993 else 8))
    rng = np.random.default_rng(0)
    xs = rng.uniform(1, 100, n)
    bh = Ah @ xs
    x0 = xs * (1 + 0.3 * rng.standard_normal(n))
    nb = np.linalg.norm(bh)
    ntiles = (n + TR - 1) // TR
    NB = max(8, (ntiles // TPB) // 8 * 8)
    rows_of = [np.arange(t * TR, min((t + 1) * TR, n)) for t in range(ntiles)]
    sched = sch.chain_schedule(mesh['edges_face1'], mesh['edges_face2'], mesh['advection_coeff'][0], n, TR, ntiles, NB, streams_per_block=1)
    lists = [[int(t) for t in sched[:, b] if t >= 0] for b in range(sched.shape[1])]
    J2r = [J2[r] for r in rows_of]

    def one_pass(xin, c2):
        xout = xin.copy()
        for lst in lists:
            y = xin.copy()                                       # rows of other lists: the pass's input
            for t in lst:
                r = rows_of[t]
                for _ in range(L):
                    y[r] = c2[r] + J2r[t] @ y
                xout[r] = y[r]
        return xout

    def stationary(maxp=400):
        x = x0.copy(); c2 = bh + J @ bh
        for p in range(1, maxp + 1):
            x = one_pass(x, c2)
            if np.linalg.norm(bh - Ah @ x) / nb < 1e-12:
                return p
        return maxp

    line = f'n={n} dt={dt:g} ||J||inf={rho:.4f} x{L}: stationary {stationary()} passes'
    for p in (1, 2, 4):
        cnt = [0]

        def prec(r):
            cnt[0] += 1
            z = np.zeros(n); c2 = r + J @ r
            for _ in range(p):
                z = one_pass(z, c2)
            return z
        M = spl.LinearOperator((n, n), matvec=prec)
        res = []
        x, info = spl.bicgstab(Ah, bh, x0=x0, rtol=1e-12, atol=0.0, M=M, maxiter=400, callback=lambda xk: res.append(0))
        rr = np.linalg.norm(bh - Ah @ x) / nb
        line += f'; BiCGSTAB p={p}: {len(res)} its = {cnt[0] * p} passes + {cnt[0]} c2 + {2 * len(res)} products (res {rr:.1e})'
        cnt[0] = 0
        x, info = spl.gmres(Ah, bh, x0=x0, rtol=1e-12, atol=0.0, M=M, restart=30, maxiter=20, callback_type='pr_norm', callback=lambda r: None)
        rr = np.linalg.norm(bh - Ah @ x) / nb
        line += f'; GMRES(30) p={p}: {cnt[0] * p} passes (res {rr:.1e})'
    print(line, flush=True)
