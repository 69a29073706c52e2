"""CPU model (round 3): lane-major tiles (tile_len cells along the flow x 64 / tile_len across), every block walking a run of
consecutive tiles of a lane in place, against the ping-pong passes; and what an IN-TILE Gauss-Seidel along the flow (the tile's
columns relaxed one after the other inside a visit) would add.  Same pessimistic round model as chain_gs_probe.py.
usage: lane_gs_probe.py [nx] [dt] [tiles per stream]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import lane_order, renumber_mesh
from oracle import cwr_oracle as orc

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
TPS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
TR = 64
mesh = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=dt, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
n = mesh['nreal'] + 1
for tile_len in (int(v) for v in os.environ.get('TILE_LENS', '4,8').split(',')):
    m = renumber_mesh(mesh, lane_order(mesh, n, tile_rows=TR, tile_len=tile_len))
    orc.derive_coefficients(m)
    lhs = orc.LHS(m); lhs.update_values(m, 0)
    A = lhs.csr().tocsr()[:n, :n]
    J = sp.identity(n, format='csr') - sp.diags(1.0 / A.diagonal()) @ A
    J.eliminate_zeros()
    J2 = (J @ J).tocsr()
    rng = np.random.default_rng(0)
    xs = rng.uniform(1, 100, n); bh = xs - J @ xs; c2 = bh + J @ bh
    x0 = xs * (1 + 0.3 * rng.standard_normal(n)); nb = np.linalg.norm(bh)
    tile = np.arange(n) // TR; ntiles = int(tile.max()) + 1
    coo = J2.tocoo(); inside = tile[coo.row] == tile[coo.col]
    Jin = sp.csr_matrix((coo.data[inside], (coo.row[inside], coo.col[inside])), shape=(n, n)).tocsr()
    Jout = sp.csr_matrix((coo.data[~inside], (coo.row[~inside], coo.col[~inside])), shape=(n, n)).tocsr()
    # flow direction along the numbering inside a lane: tiles are consecutive; odd lanes are numbered against the axis
    x_c = np.asarray(m['face_x'][:n])
    tile_x = np.bincount(tile, weights=x_c, minlength=ntiles) / np.bincount(tile, minlength=ntiles)
    # streams: runs of TPS consecutive tiles, walked downstream (ascending x)
    streams = []
    for s0 in range(0, ntiles, TPS):
        run = list(range(s0, min(s0 + TPS, ntiles)))
        if tile_x[run[-1]] < tile_x[run[0]]: run.reverse()
        streams.append(run)
    depth = max(len(s) for s in streams)
    rows_of = [np.arange(t * TR, min((t + 1) * TR, n)) for t in range(ntiles)]
    rounds = [np.concatenate([rows_of[s[i]] for s in streams if i < len(s)]) for i in range(depth)]
    # in-tile column order: quartiles of the along-flow coordinate inside every tile
    rank_in_tile = np.zeros(n, int)
    for t in range(ntiles):
        r = rows_of[t]; rank_in_tile[r] = np.argsort(np.argsort(x_c[r]))
    NCOL = 4
    col_of = rank_in_tile * NCOL // TR

    def run(label, step, maxp=300):
        x = x0.copy()
        for p in range(1, maxp + 1):
            x = step(x)
            r = np.linalg.norm(bh - (x - J @ x)) / nb
            if r < 1e-12: break
        print(f'tile {tile_len}x{TR // tile_len} {label}: {p} passes', flush=True)

    def jac(L):
        def step(x):
            g = c2 + Jout @ x; y = x
            for _ in range(L): y = g + Jin @ y
            return y
        return step

    def chains(L, gs_cols=False):
        def step(x):
            x = x.copy()
            for mrows in rounds:
                g = c2[mrows] + Jout[mrows] @ x
                y = x.copy()
                for _ in range(L):
                    if not gs_cols:
                        y[mrows] = g + (Jin[mrows] @ y)
                    else:
                        for c in range(NCOL):                       # the tile's columns one after the other, upstream first
                            sel = col_of[mrows] == c
                            y[mrows[sel]] = g[sel] + (Jin[mrows[sel]] @ y)
                x = y
            return x
        return step
    print(f'-- {n} cells, {ntiles} tiles, streams of {TPS}, dt={dt}, ||J||inf={abs(J).sum(axis=1).max():.4f}')
    for L in (int(v) for v in os.environ.get('REPS', '2,4').split(',')):
        run(f'ping-pong x{L}            ', jac(L))
        run(f'lanes in place x{L}        ', chains(L))
    run('lanes in place, column GS x1', chains(1, True))
    run('lanes in place, column GS x2', chains(2, True))
