import sys, os, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.distributed import PartitionedTransport

def hilbert_index(x, y, bits=16):
    """Hilbert curve index of integer grid points (vectorised xy2d)."""
    x = x.astype(np.int64).copy(); y = y.astype(np.int64).copy()
    d = np.zeros_like(x)
    s = 1 << (bits - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        # rotate
        swap = ry == 0
        flip = swap & (rx == 1)
        x = np.where(flip, s - 1 - x, x); y = np.where(flip, s - 1 - y, y)
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s >>= 1
    return d

def renumber(mesh, inputs3, order):
    """order[new] = old real-cell id; ghosts keep their ids."""
    n = mesh['nreal'] + 1
    ncell = len(mesh['face_x'])
    inv = np.arange(ncell); inv[order] = np.arange(n)
    full = np.arange(ncell); full[:n] = order
    m = dict(mesh)
    m['edges_face1'] = inv[mesh['edges_face1']].astype(np.int32)
    m['edges_face2'] = inv[mesh['edges_face2']].astype(np.int32)
    m['face_x'] = mesh['face_x'][full]; m['face_y'] = mesh['face_y'][full]
    m['volume'] = np.ascontiguousarray(mesh['volume'][:, full])
    return m, np.ascontiguousarray(inputs3[:, full, :])

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mesh = cw.synthetic.make_mesh(1000, 1000, 4, seed=4, dt=40.0, diffusion_coefficient=0.5)
inputs3 = cw.synthetic.boundary_input_array(mesh, K)
n = mesh['nreal'] + 1
for name in ('natural', 'hilbert', 'morton-ish blocks 16x16'):
    if name == 'natural':
        m, inp = mesh, inputs3
    else:
        x = mesh['face_x'][:n]; y = mesh['face_y'][:n]
        gx = ((x - x.min()) / (x.max() - x.min()) * 65535).astype(np.int64)
        gy = ((y - y.min()) / (y.max() - y.min()) * 65535).astype(np.int64)
        if name == 'hilbert':
            key = hilbert_index(gx, gy)
        else:
            bx, by = gx // 1049, gy // 1049          # ~16x16-cell blocks
            key = (by * 100 + bx) * 10**10 + gy * 65536 + gx
        order = np.argsort(key, kind='stable')
        m, inp = renumber(mesh, inputs3, order)
    pt = PartitionedTransport(m, inp, 0, 1)
    pt.step(0, mass_flux=False); pt.step(1, mass_flux=False)
    r = pt.step(2, mass_flux=False)
    print(f'{name}: K={K} step {r.solve_ms:.2f} ms sweeps {r.sweeps} launches {r.operator_launches}; sweep kernel {pt.engine.time_apply(2, reps=30):.1f} us', flush=True)
    os.environ['CWR_NO_SQ'] = '1'
    pt2 = PartitionedTransport(m, inp, 0, 1)
    pt2.step(0, mass_flux=False); r2 = pt2.step(1, mass_flux=False)
    print(f'   plain sweeps only: step {r2.solve_ms:.2f} ms; sweep kernel {pt2.engine.time_apply(1, reps=30):.1f} us', flush=True)
    del os.environ['CWR_NO_SQ']
