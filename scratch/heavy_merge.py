import sys; sys.path.insert(0, '/root/repo')
import numpy as np, clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
for K in (1, 12, 16):
    mesh = cw.synthetic.make_mesh(200, 100, 3, seed=7, n_merge=5000, dt=60.0, diffusion_coefficient=0.3)
    n = mesh['nreal'] + 1
    deg = np.bincount(np.r_[mesh['edges_face1'], mesh['edges_face2'][mesh['edges_face2'] < n]], minlength=n)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K)
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    r = [pt.step(t, tol=1e-12) for t in range(3)][-1]
    print(f'K={K} n={n} max faces {deg.max()} mean {deg.mean():.2f}: kernel {r.sweep_kernel} sweeps {r.sweeps} bicg {r.iterations} resid {r.max_rel_residual:.1e} {r.solve_ms:.2f} ms', flush=True)
