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
pt = PartitionedTransport(mesh, inputs3, 0, 1, renumber="hilbert")
pt.step(0, tol=1e-12, mass_flux=False); pt.step(1, tol=1e-12, mass_flux=False)
r = pt.step(2, tol=1e-12, mass_flux=False)
print("tcl rows", os.environ.get("CWR_TCL_ROWS"), "no_tcl", os.environ.get("CWR_NO_TCL"), "step ms %%.2f" %% r.solve_ms, "kernel us %%.1f" %% pt.engine.time_apply(2, reps=30), flush=True)
''' % root
for rows in ('32', '64', '96', '128', '256'):
    subprocess.run([sys.executable, '-c', code], env=dict(os.environ, CWR_TCL_ROWS=rows, CWR_VERBOSE='1'))
subprocess.run([sys.executable, '-c', code], env=dict(os.environ, CWR_NO_TCL='1'))
