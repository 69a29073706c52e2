import sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
K = int(os.environ["KK"])
mesh = cw.synthetic.make_mesh(1000, 1000, 4, seed=4, dt=40.0, diffusion_coefficient=0.5)
inputs3 = cw.synthetic.boundary_input_array(mesh, K)
pt = PartitionedTransport(mesh, inputs3, 0, 1)
pt.step(0, mass_flux=True); pt.step(1, mass_flux=True)
r = pt.step(2, mass_flux=True)
print("K", K, "sq_min_k", os.environ["CWR_SQ_MIN_K"], "kernel", r.sweep_kernel, "sweeps", r.sweeps, "launches", r.operator_launches, "step ms %%.2f" %% r.solve_ms, flush=True)
''' % root
for K in (1, 2, 4, 6):
    for mk in ('1', '64'):
        subprocess.run([sys.executable, '-c', code], env=dict(os.environ, KK=str(K), CWR_SQ_MIN_K=mk))
