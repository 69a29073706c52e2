"""CPU model: chained in-place passes that ALTERNATE between two lane tilings, the second shifted by half a lane across the flow.
The lists of one tiling are Gauss-Seidel along the flow and Jacobi-like between lanes (r04_af: making tiles longer along the flow costs
sweeps even at CFL 25/62: what limits the pass count is the coupling ACROSS lanes).  With a second tiling whose lane boundaries lie in the
middle of the first one's lanes, every cell is interior to a lane in one of two consecutive passes (alternating Schwarz).
Modelled as the engine runs a chained pass (lists of 15 tiles, tiles of one round see each other's old values, L tile-local J^2
applications).  Printed: passes to a scaled residual of 1e-12, one tiling / alternating.
usage: shifted_lanes_probe.py [nx] [dt ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp

import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import flow_axis, renumber_mesh
from clearwater_riverine_amd import schedule as sch
from oracle import cwr_oracle as orc

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 160
dts = [float(v) for v in sys.argv[2:]] or [40.0, 400.0, 1000.0]
TR, TPB, TLEN = 64, 15, 4


def lanes(mesh, n, shift):
    (ax, ay), _ = flow_axis(mesh, n)
    x = np.asarray(mesh['face_x'], dtype=np.float64)[:n]; y = np.asarray(mesh['face_y'], dtype=np.float64)[:n]
    s_along = x * ax + y * ay; q = -x * ay + y * ax
    f1 = np.asarray(mesh['edges_face1']); f2 = np.asarray(mesh['edges_face2']); real = f2 < n
    h = float(np.median(np.hypot(x[f1[real]] - x[f2[real]], y[f1[real]] - y[f2[real]])))
    width = (TR // TLEN) * h
    lane = np.floor((q - q.min() + shift * width) / width).astype(np.int64)
    key = np.where(lane & 1, -s_along, s_along)
    return np.lexsort((key, lane)).astype(np.int64)


for dt in dts:
    mesh0 = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=dt, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
    n = mesh0['nreal'] + 1
    orc.derive_coefficients(mesh0)
    lhs = orc.LHS(mesh0); lhs.update_values(mesh0, 0)
    A = lhs.csr().tocsr()[:n, :n]
    D = A.diagonal()
    J = (sp.identity(n, format='csr') - sp.diags(1.0 / D) @ A).tocsr(); J.eliminate_zeros()
    J2 = (J @ J).tocsr()
    rho = float(np.max((abs(A).sum(axis=1).A1 - np.abs(D)) / D))
    L = 2 if rho < 0.9 else (4 if rho < 0.98 else (6 if rho < 0.993 else 8))
    rng = np.random.default_rng(0)
    xs = rng.uniform(1, 100, n)
    bh = xs - J @ xs
    c2 = bh + J @ bh
    x0 = xs * (1 + 0.3 * rng.standard_normal(n))
    nb = np.linalg.norm(bh)
    ntiles = (n + TR - 1) // TR
    NB = max(8, (ntiles // TPB) // 8 * 8)
    tilings = []
    for shift in (0.0, 0.5):
        order = lanes(mesh0, n, shift)
        m = renumber_mesh(mesh0, order)
        orc.derive_coefficients(m)
        sched = sch.chain_schedule(m['edges_face1'], m['edges_face2'], m['advection_coeff'][0], n, TR, ntiles, NB, streams_per_block=1)
        rounds = [[int(t) for t in row if t >= 0] for row in sched if (row >= 0).any()]
        rows_of = [order[t * TR:min((t + 1) * TR, n)] for t in range(ntiles)]
        tilings.append((rounds, rows_of, [J2[r] for r in rows_of]))

    def run(alternate, maxp=400):
        x = x0.copy()
        for p in range(1, maxp + 1):
            rounds, rows_of, J2r = tilings[(p - 1) % 2 if alternate else 0]
            for rnd in rounds:
                xin = x.copy()
                for t in rnd:
                    r = rows_of[t]
                    y = xin.copy()
                    for _ in range(L):
                        y[r] = c2[r] + J2r[t] @ y
                    x[r] = y[r]
            if np.linalg.norm(bh - (x - J @ x)) / nb < 1e-12:
                return p
        return maxp

    print(f'n={n} dt={dt:g} ||J||inf={rho:.4f} x{L}: passes  one tiling {run(False)}  alternating shifted tilings {run(True)}', flush=True)
