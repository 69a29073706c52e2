import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mesh = cw.synthetic.make_mesh(1000, 1000, 3, seed=4, dt=40.0, diffusion_coefficient=0.5)
inputs3 = cw.synthetic.boundary_input_array(mesh, K)
pt = PartitionedTransport(mesh, inputs3, 0, 1, renumber='hilbert')
pt.step(0, tol=1e-12, mass_flux=False)
pt.step(1, tol=1e-12, mass_flux=False)
print('bytes', pt.engine.apply_bytes())
