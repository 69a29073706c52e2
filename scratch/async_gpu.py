import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
mesh = cw.synthetic.make_mesh(nx, nx, 8, seed=4, dt=40.0, diffusion_coefficient=0.5)
inputs3 = cw.synthetic.boundary_input_array(mesh, K)
pt = PartitionedTransport(mesh, inputs3, 0, 1, renumber='hilbert')
for t in range(6):
    t0 = time.time(); r = pt.step(t, tol=1e-12, mass_flux=True); el = time.time() - t0
    print(os.environ.get('CWR_LOCAL_REPS'), os.environ.get('CWR_TCL_ROWS'), 't', t, 'sweeps', r.sweeps, 'launches', r.operator_launches, 'resid %.2e' % r.max_rel_residual, 'ms %.3f' % (el * 1e3), flush=True)
x = pt.gather_state()
print('checksum', float(np.nansum(x)), float(np.nanmax(x)))
