import sys, os, subprocess, glob
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
for K in (16, 1):
    mesh = cw.synthetic.make_mesh(1000, 1000, 4, seed=4, dt=40.0, diffusion_coefficient=0.5)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K)
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    pt.step(0, tol=1e-12, mass_flux=False)
    r = pt.step(1, tol=1e-12, mass_flux=False)
    r = pt.step(2, tol=1e-12, mass_flux=False)
    print(os.path.basename(os.environ["CWR_TRANSPORT_LIB"]), "K", K, "sweeps", r.sweeps, "launches", r.operator_launches, "step ms %%.2f" %% r.solve_ms, flush=True)
''' % root
for lib in sorted(glob.glob(os.path.join(root, 'scratch', 'lib_fb*.so'))):
    env = dict(os.environ, CWR_TRANSPORT_LIB=lib)
    subprocess.run([sys.executable, '-c', code], env=env)
