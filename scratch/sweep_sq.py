import sys, os, itertools, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
K = 16
mesh = cw.synthetic.make_mesh(1000, 1000, 4, seed=4, dt=40.0, diffusion_coefficient=0.5)
inputs3 = cw.synthetic.boundary_input_array(mesh, K)
pt = PartitionedTransport(mesh, inputs3, 0, 1)
pt.step(0, tol=1e-12, mass_flux=False); pt.step(1, tol=1e-12, mass_flux=False)
r = pt.step(2, tol=1e-12, mass_flux=False)
print("tile", os.environ["CWR_TILE_ROWS"], "cap", os.environ["CWR_BLOCKS_PER_CU"], "sweeps", r.sweeps, "launches", r.operator_launches, "step ms %%.2f" %% r.solve_ms, flush=True)
''' % root
for tile, cap in itertools.product((32, 64, 128, 256), (3, 4, 6, 8)):
    env = dict(os.environ, CWR_TILE_ROWS=str(tile), CWR_BLOCKS_PER_CU=str(cap))
    subprocess.run([sys.executable, '-c', code], env=env)
