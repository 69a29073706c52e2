"""CPU experiment: tile-coloured in-place (block Gauss-Seidel) J^2 passes against the block-Jacobi passes of k_sq_tiled.
Tiles of 64 rows along the Hilbert curve are coloured so that no two tiles coupled through J^2 share a colour; a pass
processes the colours one after the other, each colour reading the rows the earlier colours have just written.
Deterministic (a colour's tiles are mutually independent) -- unlike a racy in-place update."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import hilbert_order, renumber_mesh
from oracle import cwr_oracle as orc

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
TR = int(sys.argv[3]) if len(sys.argv) > 3 else 64
mesh = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=dt, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
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
xs = rng.uniform(1, 100, n)
bh = xs - J @ xs
c2 = bh + J @ bh
J2 = (J @ J).tocsr()
x0 = xs * (1 + 0.3 * rng.standard_normal(n))
nb = np.linalg.norm(bh)

def run(label, step, maxp=400):
    x = x0.copy()
    for p in range(1, maxp + 1):
        x = step(x)
        r = np.linalg.norm(bh - (x - J @ x)) / nb
        if r < 1e-12: break
    print(f'{label}: {p} passes, resid {r:.2e}', flush=True)
    return p

run('plain J^2 pass', lambda x: c2 + J2 @ x)
tile = np.arange(n) // TR
nt = tile.max() + 1
coo = J2.tocoo()
inside = tile[coo.row] == tile[coo.col]
Jin = sp.csr_matrix((coo.data[inside], (coo.row[inside], coo.col[inside])), shape=(n, n))
Jout = sp.csr_matrix((coo.data[~inside], (coo.row[~inside], coo.col[~inside])), shape=(n, n))
# tile graph and greedy colouring in tile order
T = sp.csr_matrix((np.ones(len(coo.row)), (tile[coo.row], tile[coo.col])), shape=(nt, nt)).tocsr()
colour = -np.ones(nt, dtype=int)
for t in range(nt):
    nbc = set(colour[T.indices[T.indptr[t]:T.indptr[t + 1]]])
    c = 0
    while c in nbc: c += 1
    colour[t] = c
nc = colour.max() + 1
print(f'TR={TR}: {nt} tiles, {nc} colours, sizes {np.bincount(colour)}')
rowcol = colour[tile]
# flow-aware colour order: sort colours? (all colours are spread over the domain; order matters little)
for L in (1, 2, 3):
    def jac(x, L=L):
        g = c2 + Jout @ x
        y = x
        for _ in range(L): y = g + Jin @ y
        return y
    run(f'  block Jacobi local x{L}', jac)
    def gs(x, L=L):
        x = x.copy()
        for c in range(nc):
            m = rowcol == c
            g = (c2 + Jout @ x)[m]
            y = x.copy()
            for _ in range(L):
                y[m] = g + (Jin @ y)[m]
            x = y
        return x
    run(f'  {nc}-colour block GS local x{L}', gs)

# upper bound of what ordering can give: fully sequential block GS over the tiles sorted along the mean flow (x)
cx = np.bincount(tile, weights=mesh['face_x'][:n]) / np.bincount(tile)
seq = np.argsort(cx)
Jin_c, Jout_c = Jin.tocsr(), Jout.tocsr()
starts = np.arange(nt) * TR
for L in (2, 8):
    def sgs(x, L=L):
        x = x.copy()
        for t in seq:
            r0, r1 = starts[t], min(starts[t] + TR, n)
            g = c2[r0:r1] + Jout_c[r0:r1] @ x
            y = x[r0:r1].copy()
            Jl = Jin_c[r0:r1, r0:r1]
            for _ in range(L): y = g + Jl @ y
            x[r0:r1] = y
        return x
    run(f'  sequential downstream block GS local x{L}', sgs, maxp=60)
