"""CPU experiment: block-asynchronous J^2 passes on a PARTITIONED mesh with deep halos.
How many passes does the solve need, as a function of (reps on core tiles, reps on replayed halo tiles, passes between
two exchanges)?  Oracle operator of the synthetic mesh (test infrastructure); contiguous ranges of the Hilbert numbering."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import hilbert_order, renumber_mesh
from clearwater_riverine_amd.partition import partition_mesh
from oracle import cwr_oracle as orc

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 400
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
TR = 64
mesh = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=40.0, diffusion_coefficient=0.5)
n = mesh['nreal'] + 1
mesh = renumber_mesh(mesh, hilbert_order(mesh['face_x'], mesh['face_y'], n))
orc.derive_coefficients(mesh)
lhs = orc.LHS(mesh); lhs.update_values(mesh, 0)
A = lhs.csr().tocsr()[:n, :n]
J = (sp.identity(n, format='csr') - sp.diags(1.0 / A.diagonal()) @ A).tocsr(); J.eliminate_zeros()
rng = np.random.default_rng(0)
xs = rng.uniform(1, 100, n); bh = xs - J @ xs; c2 = bh + J @ bh
J2 = (J @ J).tocsr()
x0 = xs * (1 + 0.3 * rng.standard_normal(n))
nb = np.linalg.norm(bh)

ranks = []
for r in range(world):
    lm = partition_mesh(mesh['edges_face1'], mesh['edges_face2'], n, world, r, depth=depth)
    gl = lm.cell_global[: lm.n_rows + lm.n_halo]            # local real cells -> global ids
    # rows with a J^2 row: core + layers 1..depth-2  (partition.py numbering: core | merged layers | layer s-1 | layer s)
    nsq = lm.n_core
    if depth > 2:
        # count merged layers: rows whose every J neighbour is a computed row
        Jl = J[gl][:, gl]
        comp = np.zeros(len(gl), bool); comp[: lm.n_rows] = True
        hasrow = np.array([(Jl.indices[Jl.indptr[i]:Jl.indptr[i+1]] < lm.n_rows).all() for i in range(lm.n_rows)])
        nsq = lm.n_rows if hasrow.all() else int(np.argmin(hasrow))
    J2l = J2[gl[:nsq]][:, gl].tocoo()
    ranks.append(dict(lm=lm, gl=gl, nsq=nsq, J2l=J2l, c2=c2[gl[:nsq]]))
print(f'nx={nx} world={world} depth={depth}: core {ranks[0]["lm"].n_core}, J^2 rows {ranks[0]["nsq"]}, local {len(ranks[0]["gl"])}')

def build(rk, reps_core, reps_halo, split):
    lm, nsq, coo = rk['lm'], rk['nsq'], rk['J2l']
    tile = np.arange(nsq) // TR
    if split:                                                # tiles never straddle the core / halo boundary
        tile = np.where(np.arange(nsq) < lm.n_core, np.arange(nsq) // TR, 10**6 + (np.arange(nsq) - lm.n_core) // TR)
    nl = len(rk['gl'])
    col_tile = np.full(nl, -1); col_tile[:nsq] = tile
    inside = col_tile[coo.col] == tile[coo.row]
    rk['Jin'] = sp.csr_matrix((coo.data[inside], (coo.row[inside], coo.col[inside])), shape=(nsq, nl))
    rk['Jout'] = sp.csr_matrix((coo.data[~inside], (coo.row[~inside], coo.col[~inside])), shape=(nsq, nl))
    reps = np.where(np.arange(nsq) < lm.n_core, reps_core, reps_halo)
    rk['reps'] = reps

def run(label, reps_core, reps_halo, every, split=True, maxp=300):
    for rk in ranks:
        build(rk, reps_core, reps_halo, split)
        rk['x'] = x0[rk['gl']].copy()
    xg = x0.copy()
    for p in range(1, maxp + 1):
        if (p - 1) % every == 0:                             # exchange: every halo row <- its owner's current value
            for rk in ranks: rk['x'] = xg[rk['gl']].copy()
        for rk in ranks:
            x, nsq = rk['x'], rk['nsq']
            g = rk['c2'] + rk['Jout'] @ x
            y = x.copy()
            for rep in range(max(reps_core, reps_halo)):
                ynew = g + rk['Jin'] @ y
                act = rk['reps'] > rep
                y[:nsq][act] = ynew[act]
            rk['x'] = y
        for rk in ranks:
            lm = rk['lm']; xg[lm.lo:lm.hi] = rk['x'][: lm.n_core]
        res = np.linalg.norm(bh - (xg - J @ xg)) / nb
        if res < 1e-12: break
    print(f'{label}: {p} passes, {int(np.ceil(p / every))} exchanges, resid {res:.1e}', flush=True)

pp = depth // 2
run('exact passes, exchange every %d' % pp, 1, 1, pp)
run('core x2, halo x1, exchange every %d' % pp, 2, 1, pp)
run('core x2, halo x2, exchange every %d' % pp, 2, 2, pp)
run('core x2, halo x2, straddling tiles, exchange every %d' % pp, 2, 2, pp, split=False)
run('core x2, halo x2, exchange every %d' % max(1, pp // 2), 2, 2, max(1, pp // 2))
run('core x2, halo x1, exchange every %d' % max(1, pp // 2), 2, 1, max(1, pp // 2))
run('core x2, halo x2, exchange every pass', 2, 2, 1)
