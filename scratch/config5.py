"""BASELINE config 5 on one GPU: 2000x2000 cells, 16 constituents, a reaction step between transport steps --
device-resident (cwr_react_linear) vs a host callback through set_state/get_state (the D2H/H2D round trip)."""
import sys, os, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
K = 16
mesh = cw.synthetic.make_mesh(2000, 2000, 6, seed=5, dt=40.0, diffusion_coefficient=0.5)
inputs3 = cw.synthetic.boundary_input_array(mesh, K)
n = mesh['nreal'] + 1
pt = PartitionedTransport(mesh, inputs3, 0, 1)
eng = pt.engine
lam = np.linspace(0.0, 2e-3, K)
M = np.diag(np.exp(-lam * 40.0))
M[1, 0] = 0.01; M[0, 0] -= 0.01
eng.step(0, mass_flux=True)
t0 = time.perf_counter()
for t in range(1, 3):
    eng.react_linear(M)
    r = eng.step(t, mass_flux=True)
dev = (time.perf_counter() - t0) / 2
t0 = time.perf_counter()
for t in range(3, 5):
    c = eng.get_state()[:n]
    eng.set_state(c @ M.T)
    r = eng.step(t, mass_flux=True)
host = (time.perf_counter() - t0) / 2
print(f'config 5 (n={n}, K={K}): device reaction + step {dev*1e3:.1f} ms/step = {n*K/dev/1e6:.0f} Mcell-updates/s; '
      f'host callback (D2H {n*K*8/1e6:.0f} MB + numpy + H2D) + step {host*1e3:.1f} ms/step = {n*K/host/1e6:.0f} Mcell-updates/s; '
      f'sweeps {r.sweeps}, launches {r.operator_launches}', flush=True)
