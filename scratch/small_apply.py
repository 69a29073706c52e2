import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
for (nx, ny, K) in [(109, 27, 1), (109, 27, 12), (200, 50, 1), (200, 50, 12), (500, 200, 1), (500, 200, 16)]:
    mesh = cw.synthetic.make_mesh(nx, ny, 3, seed=1, dt=40.0)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K)
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    r = pt.step(0, mass_flux=False)
    print(f'n={mesh["nreal"]+1} K={K}: step {r.solve_ms:.3f} ms sweeps {r.sweeps} -> {1e3*r.solve_ms/max(r.sweeps,1):.1f} us/sweep incl. overheads; back-to-back apply {pt.engine.time_apply(1, reps=200):.2f} us', flush=True)
