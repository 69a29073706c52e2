"""Does the process exit cleanly with live engines / facades?  usage: exit_probe.py engine|facade|facade_closed"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
mode = sys.argv[1]
mesh = cw.synthetic.make_mesh(40, 16, 4, seed=3)
inputs3 = cw.synthetic.boundary_input_array(mesh, 2)
if mode == 'engine':
    from clearwater_riverine_amd.distributed import PartitionedTransport
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    pt.step(0)
else:
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={f'c{k}': inputs3[:, :, k].copy() for k in range(2)})
    model.update(); model.update()
    if mode == 'facade_closed':
        model.close_output()
print(mode, 'done', flush=True)
