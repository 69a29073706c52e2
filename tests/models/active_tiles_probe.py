"""Model (numpy, oracle): would SKIPPING tiles whose last update was negligible save passes' traffic?  A lane-ordered mesh of the bench generator, K = 4 distinct
constituents, one implicit step from the previous level's solution; tiles relaxed in place in lane order with two tile-local J^2 applications (the engine's
pass, modelled sequentially); per pass the share of tiles whose largest update (relative to the peak) still exceeds 1e-13 / 1e-11 / 1e-9.
Round 6 result (192 x 192): 100 % of the tiles active through pass 8, 86 % still active at pass 20 (residual 2e-9): the error decays everywhere at once --
nothing to skip.  usage: active_tiles_probe.py [nx]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'oracle'))
import numpy as np, scipy.sparse as sp
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import lane_order, renumber_mesh
import cwr_oracle as orc
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
TR = 64
steps = 6
mesh = cw.synthetic.make_mesh(nx, nx, steps, seed=4, dt=40.0, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
n = mesh['nreal'] + 1
K = 4
inp = cw.synthetic.distinct_input_array(mesh, K, seed=0)
order = lane_order(mesh, n, tile_rows=TR)
full = np.arange(len(mesh['face_x'])); full[:n] = order
mesh_r = renumber_mesh(mesh, order)
inp_r = inp[:, full, :]
orc.derive_coefficients(mesh_r)
ref = orc.OracleModel(mesh_r, {f'c{k}': inp_r[:, :, k].copy() for k in range(K)})
for s in range(steps - 1): ref.update()
t = steps - 1
lhs = orc.LHS(mesh_r); lhs.update_values(mesh_r, t)
A = lhs.csr().tocsr()[:n, :n]; D = A.diagonal()
J = sp.identity(n, format='csr') - sp.diags(1.0 / D) @ A; J.eliminate_zeros(); J2 = (J @ J).tocsr()
X0 = np.stack([ref.constituent_dict[f'c{k}'].state[t][:n] for k in range(K)], 1)
# right-hand sides of step t from the oracle
B = []
for k in range(K):
    con = ref.constituent_dict[f'c{k}']; con.b.update_values(X0[:, k], mesh_r, t); B.append(con.b.vals.copy())
B = np.stack(B, 1); bh = B / D[:, None]
XS = np.stack([sp.linalg.spsolve(A.tocsc(), B[:, k]) for k in range(K)], 1)
c2 = bh + J @ bh
tile = np.arange(n) // TR; nt = int(tile.max()) + 1
coo = J2.tocoo(); inside = tile[coo.row] == tile[coo.col]
Jin = sp.csr_matrix((coo.data[inside], (coo.row[inside], coo.col[inside])), shape=(n, n)).tocsr()
Jout = sp.csr_matrix((coo.data[~inside], (coo.row[~inside], coo.col[~inside])), shape=(n, n)).tocsr()
x = X0.copy(); peak = np.abs(XS).max(0)
print(f'{n} cells, {nt} tiles; start error {np.abs(x - XS).max(0) / peak}')
rows = [slice(tt * TR, min((tt + 1) * TR, n)) for tt in range(nt)]
for p in range(1, 21):
    act = np.zeros(nt)
    for tt in range(nt):                     # sequential Gauss-Seidel over tiles in lane order (upper bound on freshness)
        r = rows[tt]
        out = c2[r] + Jout[r] @ x
        y = x[r].copy()
        for rep in range(2):
            y = out + Jin[r][:, r] @ y
        act[tt] = np.max(np.abs(y - x[r]) / peak)
        x[r] = y
    err = np.abs(x - XS).max(0) / peak
    res = np.linalg.norm(bh + J @ x - x, axis=0) / np.linalg.norm(bh, axis=0)
    print(f'pass {p:2d}: max err {err.max():.1e}  resid {res.max():.1e}  tiles with update > 1e-13: {np.mean(act > 1e-13) * 100:5.1f} %  > 1e-11: {np.mean(act > 1e-11) * 100:5.1f} %  > 1e-9: {np.mean(act > 1e-9) * 100:5.1f} %')
    if res.max() < 1e-12: break
