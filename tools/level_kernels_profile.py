#!/usr/bin/env python3
"""Stand-alone cost of a windowed level's kernels on the 1 M-cell bench mesh (k_level_in, k_jnorm, k_note_level): a ring of 8 levels
is filled one level at a time with nothing else on the GPU; run under `rocprofv3 --kernel-trace --stats` and read the averages.
usage: level_kernels_profile.py [levels=24]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('CWR_WINDOW_EAGER', '1')
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport

T = int(sys.argv[1]) if len(sys.argv) > 1 else 24
mesh = cw.synthetic.bench_mesh(T)
inputs3 = cw.synthetic.DistinctInputs(mesh, 1, seed=4) if hasattr(cw.synthetic, 'DistinctInputs') else cw.synthetic.distinct_input_array(mesh, 1, seed=4)
pt = PartitionedTransport(mesh, inputs3, 0, 1, flow_window=8)
eng = pt.engine
eng.synchronize()
t0 = time.perf_counter()
for t in range(T - 8):
    pt.fill_window(t + 1)                 # one further level per call, as a stepping loop asks for them
    eng.synchronize()
el = (time.perf_counter() - t0) / max(1, T - 8)
print(f'LEVEL one level per call, nothing else running: {el * 1e3:.3f} ms per level (upload + kernels + synchronize)')
eng.close()
