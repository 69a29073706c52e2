import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mesh = cw.synthetic.make_mesh(1000, 1000, 3, seed=4, dt=40.0, diffusion_coefficient=0.5)
inputs3 = cw.synthetic.boundary_input_array(mesh, K)
for tile, cap in itertools.product((32, 64, 128, 256, 512), (4, 8)):
    os.environ['CWR_TILE_ROWS'] = str(tile); os.environ['CWR_BLOCKS_PER_CU'] = str(cap)
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    pt.step(0, tol=1e-12, mass_flux=False)
    us = pt.engine.time_apply(1, reps=40)
    br, bw = pt.engine.apply_bytes()
    print(f'K={K} tile_rows={tile} blocks/CU<={cap}: apply {us:.1f} us  read {br/us/1e3:.0f} GB/s  r+w {(br+bw)/us/1e3:.0f} GB/s', flush=True)
    pt.engine.close()
