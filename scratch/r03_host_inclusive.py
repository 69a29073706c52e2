"""Host-inclusive cost of the boundary (include/cwr_transport.h hands over HOST buffers): what the uploads of the inputs and
the read-outs of the results add to a step of the bench workload (1 M cells x 16).  Prints one summary line per item."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clearwater_riverine_amd import synthetic
from clearwater_riverine_amd.engine import TransportEngine
from clearwater_riverine_amd.distributed import face_to_face_distance, change_in_time

K, T = 16, 26
mesh = synthetic.bench_mesh(T, dt=40.0, diffusion_coefficient=0.5)
inputs3 = synthetic.distinct_input_array(mesh, K, seed=synthetic.BENCH_SEED)
n = mesh['nreal'] + 1
ncell = len(mesh['face_x'])
f1 = np.asarray(mesh['edges_face1'], np.int32); f2 = np.asarray(mesh['edges_face2'], np.int32)
eng = TransportEngine(f1, f2, ncell, K)
dist_e = face_to_face_distance(mesh)
dt = mesh.get('dt')
if dt is None:
    dt = change_in_time(mesh['time_seconds'] if 'time_seconds' in mesh else mesh['time'])
ff = np.ascontiguousarray(mesh['face_flow'], np.float32); ev = np.ascontiguousarray(mesh['edge_velocity'], np.float32)
vol = np.ascontiguousarray(mesh['volume'], np.float32)
def timed(f, reps=3):
    best = 1e9
    for _ in range(reps):
        eng.synchronize(); t0 = time.perf_counter(); f(); eng.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
t_load = timed(lambda: eng.load_flow_field(ff, ev, vol, dt, dist_e, float(mesh['diffusion_coefficient'])))
mb = (ff.nbytes + ev.nbytes + vol.nbytes) / 1e6
print(f'load_flow_field, {T} levels: {t_load*1e3:.1f} ms for {mb:.0f} MB of pageable numpy = {mb/1e3/t_load:.1f} GB/s -> {t_load/T*1e3:.3f} ms per level (incl. device-side derivation)')
ghost = np.ascontiguousarray(inputs3[:, n:, :])
t_b = timed(lambda: eng.load_boundary(ghost))
print(f'load_boundary: {t_b*1e3:.2f} ms for {ghost.nbytes/1e6:.1f} MB -> {t_b/T*1e3:.3f} ms per level')
x0 = np.ascontiguousarray(inputs3[0, :n, :])
t_s = timed(lambda: eng.set_state(x0))
print(f'set_state: {t_s*1e3:.2f} ms for {x0.nbytes/1e6:.0f} MB = {x0.nbytes/1e9/t_s:.1f} GB/s (pageable)')
for t in range(3):
    eng.step(t, mass_flux=True)
t_step = timed(lambda: eng.step(3, mass_flux=True), reps=1)
t_g = timed(lambda: eng.get_state())
print(f'get_state: {t_g*1e3:.2f} ms for {x0.nbytes/1e6:.0f} MB = {x0.nbytes/1e9/t_g:.1f} GB/s (pageable destination)')
# the output ring: pinned destination, asynchronous
blk = np.empty((K, n)); eng.host_register(blk)
eng.output_open(n_slots=1, with_flux=False, real_cells_only=True)
def push():
    s = eng.output_push_into(blk, None); eng.output_wait(s); eng.output_release(s)
t_p = timed(push)
print(f'output_push_into (page-locked block, state only): {t_p*1e3:.2f} ms = {blk.nbytes/1e9/t_p:.1f} GB/s')
t_f = timed(lambda: eng.get_mass_flux(), reps=2)
E = len(f1)
print(f'get_mass_flux (3 x E x K doubles = {3*E*K*8/1e6:.0f} MB): {t_f*1e3:.1f} ms')
print(f'one step here (reference numbering, no chains tuned): {t_step*1e3:.2f} ms')
