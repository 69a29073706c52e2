"""CPU experiment: tile-local repeated application of J itself (no pre-multiplied J^2) against J^2 tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import hilbert_order, renumber_mesh
from oracle import cwr_oracle as orc
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 300
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
x0 = xs * (1 + 0.3 * rng.standard_normal(n)); nb = np.linalg.norm(bh)
def split(M, TR):
    tile = np.arange(n) // TR; coo = M.tocoo(); ins = tile[coo.row] == tile[coo.col]
    return (sp.csr_matrix((coo.data[ins], (coo.row[ins], coo.col[ins])), shape=(n, n)),
            sp.csr_matrix((coo.data[~ins], (coo.row[~ins], coo.col[~ins])), shape=(n, n)))
def run(label, M, c, TR, L):
    Min, Mout = split(M, TR); x = x0.copy()
    for p in range(1, 400):
        g = c + Mout @ x; y = x
        for _ in range(L): y = g + Min @ y
        x = y
        if np.linalg.norm(bh - (x - J @ x)) / nb < 1e-12: break
    print(f'{label} TR={TR} L={L}: {p} passes', flush=True)
for TR in (64, 256):
    run('J^2', J2, c2, TR, 2); run('J^2', J2, c2, TR, 3)
    for L in (2, 4, 6, 8): run('J  ', J, bh, TR, L)
