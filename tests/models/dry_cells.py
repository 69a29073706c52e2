import sys; import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'oracle')]
import os; os.environ['CWR_NO_SMALL'] = '1'
import numpy as np, clearwater_riverine_amd as cw
import cwr_oracle as oracle
from util import oracle_run, rel_err
for K, nd in ((1, 1500), (16, 1500), (4, 3000)):
    mesh = cw.synthetic.make_mesh(120, 60, 4, seed=11, n_merge=100, n_dry=nd, dt=40.0, diffusion_coefficient=0.5)
    oracle.derive_coefficients(mesh)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K)
    names = [f'c{k}' for k in range(K)]
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)})
    ref = oracle_run(mesh, inputs3[:, :, [0]], 4)
    for _ in range(4): model.update()
    print(K, nd, 'kernel', model.last_step.sweep_kernel, 'sweeps', model.last_step.sweeps, 'rel err', rel_err(model.mesh['c0'], ref.constituent_dict['c0'].state), flush=True)
