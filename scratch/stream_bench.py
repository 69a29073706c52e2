"""Cost of the streamed output at benchmark scale: 1 M cells x 16, per step (a) no output, (b) pinned-ring copies
consumed without file writes, (c) zarr store written by the writer thread."""
import sys, os, time, threading, queue, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport
from clearwater_riverine_amd.outputs import StreamedOutput
K, nx, steps = 16, 1000, 12
mesh = cw.synthetic.make_mesh(nx, nx, steps + 2, seed=4, dt=40.0, diffusion_coefficient=0.5)
inputs3 = cw.synthetic.boundary_input_array(mesh, K)
pt = PartitionedTransport(mesh, inputs3, 0, 1, renumber='hilbert')
eng = pt.engine
gf = np.nonzero(np.asarray(mesh['edges_face2']) > mesh['nreal'])[0]
pt.set_boundary_lines([gf[0::2], gf[1::2]])
for t in range(2):
    pt.step(t, tol=1e-12, mass_flux=False)
def run(label, after_step, flux=False, mb=False):
    eng.synchronize(); t0 = time.time()
    for t in range(2, steps):
        pt.step(t, tol=1e-12, mass_flux=flux, mass_balance=mb)
        after_step(t)
    eng.synchronize(); el = (time.time() - t0) / (steps - 2)
    print(f'{label}: {el * 1e3:.3f} ms/step', flush=True)
run('no output, no mass flux', lambda t: None)
run('mass-balance ledger (2 lines of 2000 faces)', lambda t: None, mb=True)
t0 = time.time(); m = eng.domain_mass(3); print('domain_mass call: %.3f ms' % ((time.time() - t0) * 1e3))
run('per-face mass flux arrays (k_mass_flux)', lambda t: None, flux=True)
# (b) ring without file writes
eng.output_open(n_slots=3)
q = queue.Queue()
def consumer():
    while True:
        s = q.get()
        if s is None: return
        eng.output_wait(s); eng.output_release(s)
th = threading.Thread(target=consumer); th.start()
run('state -> pinned ring (no files)', lambda t: q.put(eng.output_push()))
q.put(None); th.join(); eng.output_close()
# (c) zarr store
d = tempfile.mkdtemp(dir=os.environ.get('CWR_STREAM_DIR', None))
so = StreamedOutput(eng, os.path.join(d, 'run.zarr'), [f'c{k}' for k in range(K)], steps + 2, n_slots=3)
run('state -> zarr store (%s)' % d, lambda t: so.push(t + 1))
t0 = time.time(); so.close(); print('drain at close: %.1f ms, levels written %d' % ((time.time() - t0) * 1e3, so.levels_written))
shutil.rmtree(d)
