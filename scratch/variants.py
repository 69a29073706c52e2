import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
for (nx, K) in [(1000, 16), (1000, 1), (2000, 16), (1000, 12), (1000, 4), (1000, 3)]:
    mesh = cw.synthetic.make_mesh(nx, nx, 4, seed=4 if nx == 1000 else 5, dt=40.0, diffusion_coefficient=0.5)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K)
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    pt.step(0, mass_flux=True)
    r = pt.step(1, mass_flux=True)
    r2 = pt.step(2, mass_flux=True)
    br, bw = pt.engine.apply_bytes()
    g = pt.engine.time_apply(2, reps=30, variant=0)
    a = pt.engine.time_apply(2, reps=10, variant=1)
    n = mesh['nreal'] + 1
    print(f'n={n} K={K}: step {r2.solve_ms:.2f} ms ({r2.sweeps} sweeps) = {n*K/r2.solve_ms/1e3:.0f} Mcell-updates/s; gather operator {g:.1f} us '
          f'({br/g/1e3:.0f} GB/s read, {(br+bw)/g/1e3:.0f} r+w); atomic-scatter variant {a:.1f} us ({a/g:.1f}x slower)', flush=True)
    pt.engine.close()
