"""Round-4 forensics: the library the seven faulting runs of gpurun_out/r03w_small_gs.txt used (tree of b50d8c8 + scratch/r03_small_gs.patch),
first and third case of tests/models/ohio_like.py -- the third is the first mesh above 4 096 cells."""
import sys, os
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import clearwater_riverine_amd as cw
print('library:', cw.load_library()._name, flush=True)
for (nx, ny, K) in [(109, 28, 1), (160, 50, 1)]:
    mesh = cw.synthetic.make_mesh(nx, ny, 6, seed=20100529, n_merge=0, dx=75.0, dy=75.0, depth=3.0, dt=3600.0,
                                  velocity=0.3, breathing=0.0, diffusion_coefficient=0.1, period_steps=24)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=86400.0)
    print(f'n={mesh["nreal"] + 1}: constructing + stepping', flush=True)
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    model.update()
    print(f'n={mesh["nreal"] + 1}: step done', flush=True)
