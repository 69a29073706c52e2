import sys, os, itertools, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
K = int(os.environ["KK"])
mesh = cw.synthetic.make_mesh(1000, 1000, 4, seed=4, dt=40.0, diffusion_coefficient=0.5)
inputs3 = cw.synthetic.boundary_input_array(mesh, K)
pt = PartitionedTransport(mesh, inputs3, 0, 1, renumber="hilbert")
pt.step(0, tol=1e-12, mass_flux=False); pt.step(1, tol=1e-12, mass_flux=False)
r = pt.step(2, tol=1e-12, mass_flux=False)
print("hilbert K", K, "tile", os.environ["CWR_TILE_ROWS"], "cap", os.environ["CWR_BLOCKS_PER_CU"], "step ms %%.2f" %% r.solve_ms, "kernel us %%.1f" %% pt.engine.time_apply(2, reps=30), flush=True)
''' % root
for K in (16, 1):
    for tile, cap in itertools.product((32, 64, 128) if K == 16 else (256, 512, 1024), (4, 5, 8)):
        env = dict(os.environ, CWR_TILE_ROWS=str(tile), CWR_BLOCKS_PER_CU=str(cap), KK=str(K))
        subprocess.run([sys.executable, '-c', code], env=env)
