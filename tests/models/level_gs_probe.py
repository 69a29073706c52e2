"""CPU model for DESIGN section 8 item 3: level-ordered block Gauss-Seidel J^2 passes.  Square tiles of TS x TS cells, levels
= tile columns sorted along a direction; a pass visits the levels in order, every level's tiles block-Jacobi among themselves
(L tile-local applications) and reading what the earlier levels of the SAME pass have just written.  On a GPU the levels of
successive passes would run as a time-skewed wavefront (tile (l, p) after (l-1, p) and (l+1, p-1)), deterministic.
Compared with the block-Jacobi pass of k_sq_tiled (all tiles read the previous pass)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import clearwater_riverine_amd as cw
from oracle import cwr_oracle as orc

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 240
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
TS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
mesh = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=dt, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
n = mesh['nreal'] + 1
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
X, Y = mesh['face_x'][:n], mesh['face_y'][:n]
tx = np.minimum((X - X.min()) / (X.max() - X.min() + 1e-9) * (nx / TS), nx // TS - 1).astype(int)
ty = np.minimum((Y - Y.min()) / (Y.max() - Y.min() + 1e-9) * (nx / TS), nx // TS - 1).astype(int)
tile = ty * (nx // TS) + tx
coo = J2.tocoo()
inside = tile[coo.row] == tile[coo.col]
Jin = sp.csr_matrix((coo.data[inside], (coo.row[inside], coo.col[inside])), shape=(n, n))
Jout = sp.csr_matrix((coo.data[~inside], (coo.row[~inside], coo.col[~inside])), shape=(n, n)).tocsr()

def run(label, step, maxp=200):
    x = x0.copy()
    for p in range(1, maxp + 1):
        x = step(x)
        r = np.linalg.norm(bh - (x - J @ x)) / nb
        if r < 1e-12: break
    print(f'{label}: {p} passes, resid {r:.2e}', flush=True)
    return p

print(f'{n} cells, tiles of {TS}x{TS}, {nx // TS} levels per direction, CFL set by dt={dt}')
for L in (1, 2, 3):
    def jac(x, L=L):
        g = c2 + Jout @ x; y = x
        for _ in range(L): y = g + Jin @ y
        return y
    run(f'block Jacobi (all tiles read the previous pass), local x{L}', jac)
    for name, lev in (('downstream (+x)', tx), ('upstream (-x)', tx.max() - tx), ('across (+y)', ty)):
        masks = [np.nonzero(lev == l)[0] for l in range(lev.max() + 1)]
        def lgs(x, L=L, masks=masks):
            x = x.copy()
            for m in masks:
                g = c2[m] + Jout[m] @ x
                y = x.copy()
                for _ in range(L):
                    y[m] = g + (Jin[m] @ y)
                x = y
            return x
        run(f'  level GS {name}, local x{L}', lgs, maxp=120)
