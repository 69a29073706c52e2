"""CPU experiment: how much closer to the solution of step t is a time-extrapolated initial guess than the previous level?
Sweeps saved = log(err(c_t) / err(guess)) / log(1 / rho), rho = the Jacobi contraction per sweep.  Bench-like mesh (5 % merged
cells, CFL ~ 2.5 at dt = 40 s), the four constituent families of synthetic.distinct_input_array."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import clearwater_riverine_amd as cw
from oracle import cwr_oracle as orc

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEPS = 8
mesh = cw.synthetic.make_mesh(nx, nx, STEPS + 2, seed=4, dt=40.0, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
n = mesh['nreal'] + 1
K = 8
inp = cw.synthetic.distinct_input_array(mesh, K, seed=4)
orc.derive_coefficients(mesh)
names = [f'c{k}' for k in range(K)]
model = orc.OracleModel(mesh, {nm: inp[:, :, k].copy() for k, nm in enumerate(names)})
hist = [np.stack([model.constituent_dict[nm].state[0][:n] for nm in names], axis=1)]
for t in range(STEPS):
    model.update()
    hist.append(np.stack([model.constituent_dict[nm].state[t + 1][:n] for nm in names], axis=1))
# contraction of the Jacobi sweep at level 0
lhs = orc.LHS(mesh); lhs.update_values(mesh, 0)
A = lhs.csr().tocsr()[:n, :n]
D = A.diagonal()
J = sp.identity(n, format='csr') - sp.diags(1.0 / D) @ A
v = np.random.default_rng(0).random(n)
for _ in range(200):
    w = J @ v; rho = np.linalg.norm(w) / np.linalg.norm(v); v = w / np.linalg.norm(w)
print(f'n = {n}, Jacobi contraction per sweep ~ {rho:.3f}')
for t in range(2, STEPS):
    sol = hist[t + 1]
    g0 = hist[t]
    g1 = 2 * hist[t] - hist[t - 1]
    g2 = 3 * hist[t] - 3 * hist[t - 1] + hist[t - 2] if t >= 2 else g1
    row = []
    for k in range(K):
        e0 = np.linalg.norm(sol[:, k] - g0[:, k]); e1 = np.linalg.norm(sol[:, k] - g1[:, k]); e2 = np.linalg.norm(sol[:, k] - g2[:, k])
        row.append(f'{np.log(e0 / e1) / np.log(1 / rho):+.1f}/{np.log(e0 / e2) / np.log(1 / rho):+.1f}')
    print(f'step {t}: sweeps saved per constituent, linear/quadratic extrapolation:', ' '.join(row))
