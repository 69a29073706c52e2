"""BASELINE config 5 on one GPU: the 4 M-cell merged mesh (synthetic.bench_mesh(scale=2)), 16 distinct constituents, a
reaction step between transport steps -- device-resident (cwr_react_linear) vs a host callback through
get_state / set_state (the D2H / H2D round trip of the reference's update_concentration contract)."""
import importlib.util
import os
import sys
import time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
spec = importlib.util.spec_from_file_location('large', os.path.join(root, 'tests', 'golden', 'make_expected_large.py'))
large = importlib.util.module_from_spec(spec); spec.loader.exec_module(large)
K, STEPS, WARM = 16, 8, 5
mesh = cw.synthetic.bench_mesh(2 * STEPS + WARM + 1, scale=2)
inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=cw.synthetic.BENCH_SEED + 1)
n = mesh['nreal'] + 1
pt = PartitionedTransport(mesh, inputs3, 0, 1)
eng = pt.engine
M = large.reaction_matrix(K, 40.0)
# (warm-up WITH the reaction: every batch shape the timed steps take has had its graph built -- a first use costs ~10 ms at 4 M cells,
# and with two plain warm-up steps it fell into the timed region: 13.3 instead of 12.2 ms per step, round 5)
for t in range(WARM):
    if t >= 2:
        eng.react_linear(M)
    eng.step(t, mass_flux=True)
eng.synchronize()
t0 = time.perf_counter()
for t in range(WARM, WARM + STEPS):
    eng.react_linear(M)
    r = eng.step(t, mass_flux=True)
eng.synchronize()
dev = (time.perf_counter() - t0) / STEPS
t0 = time.perf_counter()
for t in range(WARM + STEPS, WARM + STEPS + 3):
    c = eng.get_state()[:n]
    eng.set_state(c @ M.T)
    r2 = eng.step(t, mass_flux=True)
eng.synchronize()
host = (time.perf_counter() - t0) / 3
print(f'config 5 (n={n}, K={K}, merged mesh): device reaction + step {dev*1e3:.2f} ms/step = {n*K/dev/1e6:.0f} Mcell-updates/s; '
      f'host callback (D2H {n*K*8/1e6:.0f} MB + numpy + H2D) + step {host*1e3:.1f} ms/step = {n*K/host/1e6:.0f} Mcell-updates/s; '
      f'sweeps {r.sweeps}, launches {r.operator_launches}', flush=True)
