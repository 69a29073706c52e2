"""Target of the PMC passes (rocprofv3 --pmc ... -- python3 tools/pmc_target.py <mesh> <K>): two steps of the bench
workload, nothing else.  bench.py runs it as a child BEFORE it touches the GPU itself; profiles/*pmc* come from it too."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport

which = sys.argv[1] if len(sys.argv) > 1 else 'merged'
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
if which == 'quad':
    mesh = cw.synthetic.make_mesh(1000, 1000, 3, seed=4, dt=40.0, diffusion_coefficient=0.5)
else:
    mesh = cw.synthetic.bench_mesh(3)
inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=cw.synthetic.BENCH_SEED)
pt = PartitionedTransport(mesh, inputs3, 0, 1, renumber='hilbert')
pt.step(0, tol=1e-12, mass_flux=False)
pt.step(1, tol=1e-12, mass_flux=False)
print('bytes', pt.engine.apply_bytes())
