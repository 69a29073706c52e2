"""Model (numpy, oracle): what a time-extrapolated initial guess x0 = 2 x_t - x_{t-1} would be worth to the sweeps, on a small
copy of the bench workload (same generator, same CFL and inputs).  Prints, per step, the error of both guesses against the
converged solution (max norm and the scaled 2-norm the acceptance rule uses)."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'oracle'))
import numpy as np
import clearwater_riverine_amd as cw
import cwr_oracle as oracle
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 160
K = 8
steps = 24
mesh = cw.synthetic.make_mesh(nx, nx, steps + 1, seed=4, n_merge=nx * nx // 20, dt=40.0, diffusion_coefficient=0.5)
inp = cw.synthetic.distinct_input_array(mesh, K, seed=0)
oracle.derive_coefficients(mesh)
names = [f'c{k}' for k in range(K)]
ref = oracle.OracleModel(mesh, {nm: inp[:, :, k].copy() for k, nm in enumerate(names)})
n = mesh['nreal'] + 1
X = [np.stack([ref.constituent_dict[nm].state[0][:n] for nm in names], 1)]
for s in range(steps):
    ref.update()
    X.append(np.stack([ref.constituent_dict[nm].state[s + 1][:n] for nm in names], 1))
for s in range(2, steps):
    xs = X[s + 1]
    g0 = X[s]; g1 = 2 * X[s] - X[s - 1]; g2 = 3 * X[s] - 3 * X[s - 1] + X[s - 2]
    den = np.abs(xs).max(0)
    e0 = np.abs(xs - g0).max(0) / den; e1 = np.abs(xs - g1).max(0) / den; e2 = np.abs(xs - g2).max(0) / den
    print(f'step {s}: max-norm error of the guess, per constituent: x_t ' + ' '.join(f'{v:.1e}' for v in e0))
    print(f'          linear   ' + ' '.join(f'{v:.1e}' for v in e1))
    print(f'          quadratic' + ' '.join(f'{v:.1e}' for v in e2))
