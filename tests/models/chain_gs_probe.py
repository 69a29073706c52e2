"""CPU model of the flow-ordered IN-PLACE block-asynchronous pass (DESIGN "chained passes").

The GPU pass is a persistent grid: NB resident blocks, each walking its own list of 64-row tiles (Hilbert order).  Today the
pass ping-pongs between two vectors (block Jacobi between tiles, L tile-local J^2 applications).  Modelled here:

  jacobi      today's pass (every tile reads the previous pass)
  inplace     ONE vector, today's tile order: the tiles a block visits in round i see what rounds < i wrote
  chains      ONE vector, every block walks a CHAIN of tiles linked along the flow (tile -> the neighbour tile that takes
              most of its outflow, kept when that neighbour's largest inflow comes from this tile), chains cut / packed into
              NB lists of equal length: a tile sees its upstream neighbour of the same chain already relaxed
  reverse     the chains walked against the flow (the worst case: must not be worse than jacobi)

Rounds are modelled pessimistically: tiles relaxed in the same round see each other's OLD values.
usage: chain_gs_probe.py [nx] [dt] [tiles per block]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp

import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import hilbert_order, renumber_mesh
from oracle import cwr_oracle as orc

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
TPB = int(sys.argv[3]) if len(sys.argv) > 3 else 15          # tiles per block (1 M cells: 15 625 tiles / 1 024 blocks)
TR = 64
mesh = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=dt, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
n = mesh['nreal'] + 1
mesh = renumber_mesh(mesh, hilbert_order(mesh['face_x'], mesh['face_y'], n))
orc.derive_coefficients(mesh)
lhs = orc.LHS(mesh); lhs.update_values(mesh, 0)
A = lhs.csr().tocsr()[:n, :n]
D = A.diagonal()
J = sp.identity(n, format='csr') - sp.diags(1.0 / D) @ A
J.eliminate_zeros()
J2 = (J @ J).tocsr()
rng = np.random.default_rng(0)
xs = rng.uniform(1, 100, n)
bh = xs - J @ xs
c2 = bh + J @ bh
x0 = xs * (1 + 0.3 * rng.standard_normal(n))
nb = np.linalg.norm(bh)
tile = np.arange(n) // TR
ntiles = int(tile.max()) + 1
NB = max(1, ntiles // TPB)
coo = J2.tocoo()
inside = tile[coo.row] == tile[coo.col]
Jin = sp.csr_matrix((coo.data[inside], (coo.row[inside], coo.col[inside])), shape=(n, n)).tocsr()
Jout = sp.csr_matrix((coo.data[~inside], (coo.row[~inside], coo.col[~inside])), shape=(n, n)).tocsr()
rows_of = [np.arange(t * TR, min((t + 1) * TR, n)) for t in range(ntiles)]
print(f'{n} cells, {ntiles} tiles of {TR} rows, {NB} blocks x {ntiles / NB:.1f} tiles, dt={dt}, ||J||_inf={abs(J).sum(axis=1).max():.4f}')


def chains_from_flow(reverse=False):
    """Tile chains along the flow of level 0: flux[t -> u] = sum of the face flows from cells of t into cells of u."""
    f1 = np.asarray(mesh['edges_face1']); f2 = np.asarray(mesh['edges_face2'])
    a = np.asarray(mesh['advection_coeff'][0], dtype=np.float64)
    real = f2 < n
    src = np.where(a > 0, f1, f2)[real]; dst = np.where(a > 0, f2, f1)[real]
    w = np.abs(a[real])
    ts, td = tile[src], tile[dst]
    m = ts != td
    F = sp.coo_matrix((w[m], (ts[m], td[m])), shape=(ntiles, ntiles)).tocsr()
    if reverse:
        F = F.T.tocsr()
    best_dn = np.full(ntiles, -1); best_up = np.full(ntiles, -1)
    Fc = F.tocsc()
    for t in range(ntiles):
        r = F.getrow(t)
        if r.nnz: best_dn[t] = r.indices[np.argmax(r.data)]
        c = Fc.getcol(t)
        if c.nnz: best_up[t] = c.indices[np.argmax(c.data)]
    nxt = np.full(ntiles, -1)
    for t in range(ntiles):
        d = best_dn[t]
        if d >= 0 and best_up[d] == t: nxt[t] = d
    has_prev = np.zeros(ntiles, bool); has_prev[nxt[nxt >= 0]] = True
    chains, seen = [], np.zeros(ntiles, bool)
    for start in list(np.nonzero(~has_prev)[0]) + list(range(ntiles)):     # heads first, then whatever sits on a cycle
        if seen[start]: continue
        ch, t = [], start
        while t >= 0 and not seen[t]:
            seen[t] = True; ch.append(t); t = nxt[t]
        chains.append(ch)
    return chains


def pack(chains, nblocks):
    """Cut the chains into pieces of <= ceil(ntiles / nblocks) tiles and deal the pieces to the blocks, longest first,
    always to the least-loaded block: lists[b] = the tiles block b walks, in order."""
    cap = -(-ntiles // nblocks)
    pieces = []
    for ch in chains:
        for i in range(0, len(ch), cap): pieces.append(ch[i:i + cap])
    pieces.sort(key=len, reverse=True)
    lists = [[] for _ in range(nblocks)]
    load = np.zeros(nblocks, int)
    for p in pieces:
        b = int(np.argmin(load))
        lists[b] += p; load[b] += len(p)
    return lists


def rounds_of(lists):
    depth = max(len(l) for l in lists)
    return [np.concatenate([rows_of[l[i]] for l in lists if i < len(l)]) for i in range(depth)]


def run(label, step, maxp=300):
    x = x0.copy()
    for p in range(1, maxp + 1):
        x = step(x)
        r = np.linalg.norm(bh - (x - J @ x)) / nb
        if r < 1e-12: break
    print(f'{label}: {p} passes, resid {r:.2e}', flush=True)
    return p


def inplace_step(rounds, L):
    def step(x):
        x = x.copy()
        for m in rounds:
            g = c2[m] + Jout[m] @ x
            y = x.copy()
            for _ in range(L): y[m] = g + (Jin[m] @ y)
            x = y
        return x
    return step


ch = chains_from_flow()
lens = np.array([len(c) for c in ch])
print(f'chains: {len(ch)}, mean length {lens.mean():.1f}, longest {lens.max()}, tiles in chains of >= 4: {lens[lens >= 4].sum() / ntiles:.2f}')
order_now = [list(range(b, ntiles, NB)) for b in range(NB)]            # today's static share: tiles b, b + NB, ...
for L in (2, 3, 4, 6):
    def jac(x, L=L):
        g = c2 + Jout @ x; y = x
        for _ in range(L): y = g + Jin @ y
        return y
    pj = run(f'L={L} jacobi (ping-pong)          ', jac)
    run(f'L={L} inplace, today\'s tile order ', inplace_step(rounds_of(order_now), L))
    run(f'L={L} chains along the flow       ', inplace_step(rounds_of(pack(ch, NB)), L))
    run(f'L={L} chains AGAINST the flow     ', inplace_step(rounds_of(pack(chains_from_flow(True), NB)), L))
