"""Does the process exit cleanly with live engines / facades?
usage: exit_probe.py engine|engine_global|facade|facade_closed|facade_stream   (the product as it is; tests/test_gpu_exit.py)
       exit_probe.py facade_r03|facade_leak                                    (forensics, round 4: the facade of 87ad206 rebuilt
           by removing the atexit hook -- r03: __del__ releases the page locks during interpreter finalisation; leak: nobody does)"""
import atexit, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
mode = sys.argv[1]
mesh = cw.synthetic.make_mesh(40, 16, 4, seed=3)
inputs3 = cw.synthetic.boundary_input_array(mesh, 2)
if mode.startswith('engine'):
    from clearwater_riverine_amd.distributed import PartitionedTransport
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    pt.step(0)
    if mode == 'engine_global':
        KEEP = [pt, pt.engine]                      # alive until the module dict is cleared at finalisation
else:
    import clearwater_riverine_amd.model as model_mod
    kw = {}
    if mode == 'facade_stream':
        import tempfile
        kw = dict(output_store=os.path.join(tempfile.mkdtemp(), 'run.zarr'))
    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={f'c{k}': inputs3[:, :, k].copy() for k in range(2)}, **kw)
    model.update(); model.update()
    if mode == 'facade_closed':
        model.close_output()
    if mode in ('facade_r03', 'facade_leak'):
        atexit.unregister(model_mod._close_all_models)
        if mode == 'facade_r03':
            model_mod.ClearwaterRiverine.__del__ = lambda self: self.close_output()
        else:
            model_mod.ClearwaterRiverine.__del__ = lambda self: None
print(mode, 'done', flush=True)
