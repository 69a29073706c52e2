"""Sweeps and operator launches per step on the bench mesh (checks per step = batches), for a margin on the predicted sweeps."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
K = int(sys.argv[1]); STEPS = 40
mesh = cw.synthetic.bench_mesh(STEPS + 2)
inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=cw.synthetic.BENCH_SEED)
pt = PartitionedTransport(mesh, inputs3, 0, 1)
eng = pt.engine
rows = []
for t in range(STEPS):
    if t == 8:
        eng.synchronize(); t0 = time.perf_counter()
    r = eng.step(t, mass_flux=True)
    rows.append((r.sweeps, r.operator_launches))
eng.synchronize()
ms = (time.perf_counter() - t0) / (STEPS - 8) * 1e3
two = sum(1 for s, l in rows[8:] if l > (s + 1) // 2 + 1)
print(f"K={K} margin={os.environ.get('CWR_SWEEP_MARGIN','0')} two_closing={os.environ.get('CWR_TWO_CLOSING','0')}: {ms:.3f} ms/step over {STEPS-8} steps; steps with more than one batch: {two}; (sweeps, launches): {rows[8:]}", flush=True)
