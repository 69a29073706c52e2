"""Round 3: mid-size meshes -- is it worth shrinking the persistent grid so that every block gets a list long enough to chain?
usage: r03_mid.py <nx> <K> [grid caps ...]   (0 = the engine's default grid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
nx, K = int(sys.argv[1]), int(sys.argv[2])
caps = [int(v) for v in sys.argv[3:]] or [0]
steps, warm = 12, 3
mesh = cw.synthetic.make_mesh(nx, nx, warm + steps + 1, seed=4, dt=float(os.environ.get("MID_DT", "40")), diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=4)
for cap in caps:
    for chains in (False, True):
        os.environ.pop('CWR_TCL_GRID', None); os.environ.pop('CWR_NO_CHAINS', None)
        if cap: os.environ['CWR_TCL_GRID'] = str(cap)
        if not chains: os.environ['CWR_NO_CHAINS'] = '1'
        pt = PartitionedTransport(mesh, inputs3, 0, 1)
        for t in range(warm): pt.step(t, tol=1e-12)
        pt.engine.synchronize(); t0 = time.perf_counter(); sw = []
        for t in range(warm, warm + steps): sw.append(pt.step(t, tol=1e-12).sweeps)
        pt.engine.synchronize(); el = (time.perf_counter() - t0) / steps
        ok, ntiles, grid, TR = pt.engine.tiling_info()
        sched = pt.engine.get_tile_schedule()[0]
        print(f'{mesh["nreal"] + 1} cells x {K}: grid {grid} ({ntiles / grid:.1f} tiles per block) chains {"on " if sched is not None else "off"}: {el * 1e3:.3f} ms/step, sweeps {min(sw)}-{max(sw)}', flush=True)
        pt.engine.close()
