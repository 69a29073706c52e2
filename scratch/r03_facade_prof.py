import sys, os, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import clearwater_riverine_amd as cw
nx, ny, nm, K, steps = 109, 28, 109, 12, 200
mesh = cw.synthetic.make_mesh(nx, ny, steps, seed=20100529, n_merge=nm, dx=75.0, dy=75.0, depth=3.0, dt=3600.0, velocity=0.3, breathing=0.0, diffusion_coefficient=0.1, period_steps=24)
inputs3 = cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=86400.0)
names = [f'c{k}' for k in range(K)]
model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={n_: inputs3[:, :, k].copy() for k, n_ in enumerate(names)})
eng = model.engine
acc = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
for nme in ('step', 'output_push', 'output_wait', 'output_release', 'domain_mass'):
    wrap(eng, nme)
model.update(); model.update()
acc.clear()
t0 = time.perf_counter()
N = 150
for _ in range(N):
    model.update()
tot = time.perf_counter() - t0
print(f'facade {tot / N * 1e3:.3f} ms/step; ' + ', '.join(f'{k} {v / N * 1e3:.3f}' for k, v in acc.items()) + f'; python rest {(tot - sum(acc.values())) / N * 1e3:.3f}')
