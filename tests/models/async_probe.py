"""CPU experiment: how many passes does a tile-local repeated J^2 application (block-asynchronous Jacobi)
need against plain J^2 passes?  Uses the oracle operator of the synthetic mesh (test infrastructure)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import hilbert_order, renumber_mesh
from oracle import cwr_oracle as orc

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
mesh = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=dt, diffusion_coefficient=0.5)
n = mesh['nreal'] + 1
order = hilbert_order(mesh['face_x'], mesh['face_y'], n)
mesh = renumber_mesh(mesh, order)
orc.derive_coefficients(mesh)
lhs = orc.LHS(mesh); lhs.update_values(mesh, 0)
A = lhs.csr().tocsr()[:n, :n]
D = A.diagonal()
J = sp.identity(n, format='csr') - sp.diags(1.0 / D) @ A
J.eliminate_zeros()
rng = np.random.default_rng(0)
xs = rng.uniform(1, 100, n)          # "true" solution
bh = xs - J @ xs                       # scaled rhs: (I-J) xs
c2 = bh + J @ bh
J2 = (J @ J).tocsr()
x0 = xs * (1 + 0.3 * rng.standard_normal(n))

def run(label, step, maxp=400):
    x = x0.copy(); nb = np.linalg.norm(bh)
    for p in range(1, maxp + 1):
        x = step(x)
        r = np.linalg.norm(bh - (x - J @ x)) / nb
        if r < 1e-12: break
    print(f'{label}: {p} passes, resid {r:.2e}', flush=True)
    return p

base = run('plain J^2 pass', lambda x: c2 + J2 @ x)

for TR in (128, 256, 512, 1024):
    tile = np.arange(n) // TR
    coo = J2.tocoo()
    inside = tile[coo.row] == tile[coo.col]
    Jin = sp.csr_matrix((coo.data[inside], (coo.row[inside], coo.col[inside])), shape=(n, n))
    Jout = sp.csr_matrix((coo.data[~inside], (coo.row[~inside], coo.col[~inside])), shape=(n, n))
    print(f'TR={TR}: inside-tile fraction of J^2 weight {Jin.sum() / J2.sum():.3f}')
    for L in (2, 3, 4):
        def step(x, L=L):
            g = c2 + Jout @ x          # stale part, frozen during the local applications
            y = x
            for _ in range(L):
                y = g + Jin @ y
            return y
        p = run(f'  TR={TR} local x{L}', step)
