import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
os.environ['CWR_NO_SMALL'] = '1'; os.environ['CWR_TWO_CLOSING'] = '1'; os.environ['CWR_LOCAL_REPS'] = '1'
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
mesh = cw.synthetic.make_mesh(48, 20, 4, seed=21, n_merge=60, shuffle_window=16, n_dry=2)
inputs3 = cw.synthetic.boundary_input_array(mesh, 3)
pt = PartitionedTransport(mesh, inputs3, 0, 1)
for t in range(3):
    r = pt.step(t, tol=1e-12, mass_flux=True, solver='jacobi')
    print(os.environ.get('CWR_TRANSPORT_LIB', 'in-tree'), t, r.sweeps, r.operator_launches, r.max_rel_residual, float(np.nansum(pt.gather_state())))
np.save(sys.argv[1], pt.gather_state())
