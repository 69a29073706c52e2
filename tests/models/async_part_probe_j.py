"""CPU experiment: tile-local repeated application of J (not J^2) on a PARTITIONED mesh with deep halos:
passes needed vs (local applications L, passes between exchanges)."""
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
TR = int(sys.argv[4]) if len(sys.argv) > 4 else 256
mesh = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=40.0, diffusion_coefficient=0.5)
n = mesh['nreal'] + 1
mesh = renumber_mesh(mesh, hilbert_order(mesh['face_x'], mesh['face_y'], n))
orc.derive_coefficients(mesh)
lhs = orc.LHS(mesh); lhs.update_values(mesh, 0)
A = lhs.csr().tocsr()[:n, :n]
J = (sp.identity(n, format='csr') - sp.diags(1.0 / A.diagonal()) @ A).tocsr(); J.eliminate_zeros()
rng = np.random.default_rng(0)
xs = rng.uniform(1, 100, n); bh = xs - J @ xs
x0 = xs * (1 + 0.3 * rng.standard_normal(n)); nb = np.linalg.norm(bh)
ranks = []
for r in range(world):
    lm = partition_mesh(mesh['edges_face1'], mesh['edges_face2'], n, world, r, depth=depth)
    gl = lm.cell_global[: lm.n_rows + lm.n_halo]
    nr = lm.n_rows                                           # computed rows: core + layers 1..depth-1
    Jl = J[gl[:nr]][:, gl].tocoo()
    tile = np.arange(nr) // TR
    col_tile = np.full(len(gl), -1); col_tile[:nr] = tile
    ins = col_tile[Jl.col] == tile[Jl.row]
    ranks.append(dict(lm=lm, gl=gl, nr=nr, bh=bh[gl[:nr]],
                      Jin=sp.csr_matrix((Jl.data[ins], (Jl.row[ins], Jl.col[ins])), shape=(nr, len(gl))),
                      Jout=sp.csr_matrix((Jl.data[~ins], (Jl.row[~ins], Jl.col[~ins])), shape=(nr, len(gl)))))
print(f'nx={nx} world={world} depth={depth} TR={TR}: core {ranks[0]["lm"].n_core}, computed {ranks[0]["nr"]}')
def run(label, L, every, maxp=400):
    for rk in ranks: rk['x'] = x0[rk['gl']].copy()
    xg = x0.copy()
    for p in range(1, maxp + 1):
        if (p - 1) % every == 0:
            for rk in ranks: rk['x'] = xg[rk['gl']].copy()
        for rk in ranks:
            x, nr = rk['x'], rk['nr']
            g = rk['bh'] + rk['Jout'] @ x
            y = x.copy()
            for _ in range(L): y[:nr] = g + rk['Jin'] @ y
            rk['x'] = y
        for rk in ranks:
            lm = rk['lm']; xg[lm.lo:lm.hi] = rk['x'][: lm.n_core]
        res = np.linalg.norm(bh - (xg - J @ xg)) / nb
        if res < 1e-12: break
    print(f'{label}: {p} passes, {int(np.ceil(p / every))} exchanges', flush=True)
if world == 1:
    for L in (1, 2, 4, 6): run(f'single L={L}', L, 10**9)
else:
    for L in (1, 4):
        for every in (depth, depth // 2, max(1, depth // 4)):
            run(f'L={L} exchange every {every}', L, every)
