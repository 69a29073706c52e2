"""CPU model: chained in-place passes with the LANES taken in two colours (even lanes first, then odd lanes).

Within a pass of the engine all lists run concurrently: along a lane a tile sees its predecessor's fresh results (Gauss-Seidel), but
ACROSS lanes the coupling (diffusion, the cross-flow component of the field) is block-Jacobi -- a tile sees its side neighbours as the
previous pass left them.  Modelled here: the same chains, but every pass walks the tiles of the even lanes first and those of the odd
lanes second (two launches of half the tiles each on the GPU), so that an odd lane sees both its neighbours already relaxed.
usage: lane_colour_probe.py [nx] [dt ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp

import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import lane_order, renumber_mesh
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
    J = sp.identity(n, format='csr') - sp.diags(1.0 / D) @ A
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
    coo = J2.tocoo()
    inside = tile[coo.row] == tile[coo.col]
    Jin = sp.csr_matrix((coo.data[inside], (coo.row[inside], coo.col[inside])), shape=(n, n)).tocsr()
    Jout = sp.csr_matrix((coo.data[~inside], (coo.row[~inside], coo.col[~inside])), shape=(n, n)).tocsr()
    rows_of = [np.arange(t * TR, min((t + 1) * TR, n)) for t in range(ntiles)]
    us, ud, w = sch.tile_links(mesh['edges_face1'], mesh['edges_face2'], mesh['advection_coeff'][0], n, TR, ntiles)
    chains = sch.chains(us, ud, w, ntiles)
    # lane of a tile: lanes are 16 cells wide (TR / 4) in the lane-major numbering; the cross-flow coordinate of the tile's centre
    yc = np.array([np.asarray(mesh['face_y'])[rows_of[t]].mean() for t in range(ntiles)])
    lane = np.floor((yc - yc.min()) / (16 * 10.0) + 0.25).astype(int)

    def rounds_for(subset_chains, nblocks):
        s = sch.schedule(subset_chains, sum(len(c) for c in subset_chains), nblocks, streams_per_block=1) if False else None
        seq = np.concatenate([np.asarray(c) for c in subset_chains])
        bounds = (np.arange(nblocks + 1) * len(seq)) // nblocks
        lists = [seq[bounds[b]:bounds[b + 1]] for b in range(nblocks)]
        depth = max(len(l) for l in lists)
        return [np.concatenate([rows_of[l[i]] for l in lists if i < len(l)]) for i in range(depth)]

    def run(rounds_list, maxp=300):
        x = x0.copy()
        for p in range(1, maxp + 1):
            for rounds in rounds_list:
                for m in rounds:
                    g = c2[m] + Jout[m] @ x
                    y = x.copy()
                    for _ in range(L):
                        y[m] = g + (Jin[m] @ y)
                    x = y
            if np.linalg.norm(bh - (x - J @ x)) / nb < 1e-12:
                return p
        return maxp

    allr = rounds_for(chains, NB)
    even = [c for c in chains if lane[c[0]] % 2 == 0]
    odd = [c for c in chains if lane[c[0]] % 2 == 1]
    two = [rounds_for(even, NB), rounds_for(odd, NB)]
    three = [rounds_for([c for c in chains if lane[c[0]] % 3 == q], NB) for q in range(3)]
    print(f'n={n} dt={dt:g} ||J||_inf={rho:.4f} x{L}: {len(chains)} chains, {lane.max() + 1} lanes; one colour (today) {run([allr])} passes; '
          f'two colours {run(two)}; three colours {run(three)}', flush=True)
