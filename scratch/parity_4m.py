"""Where do the HIP path and the oracle part ways on the 4 M-cell mesh?  Operator, right-hand side and one solve, each
against the oracle's vectorised per-cell forms (numpy on the host; the solve through its true residual)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'oracle'))
import numpy as np
import cwr_oracle as oracle
import clearwater_riverine_amd as cw
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = 2
mesh = cw.synthetic.bench_mesh(3, scale=scale)
oracle.derive_coefficients(mesh)
inputs3 = cw.synthetic.distinct_input_array(mesh, 16, seed=cw.synthetic.BENCH_SEED + 1)[:, :, [5, 7]].copy()
n = mesh['nreal'] + 1
ncell = len(mesh['face_x'])
from clearwater_riverine_amd.ordering import hilbert_order
for order in (None, hilbert_order(mesh['face_x'], mesh['face_y'], n)):
    eng = cw.TransportEngine(mesh['edges_face1'], mesh['edges_face2'], ncell, K, cell_order=order)
    eng.load_flow_field(mesh['face_flow'], mesh['edge_velocity'], mesh['volume'], mesh['dt'], mesh['face_to_face_dist'], mesh['diffusion_coefficient'])
    eng.load_boundary(inputs3[:, n:, :])
    adv, dif = eng.get_coefficients(0)
    print('order', 'hilbert' if order is not None else 'none', 'coeff equal', np.array_equal(adv, mesh['advection_coeff'][0]), np.array_equal(dif, mesh['coeff_to_diffusion'][0]))
    x = np.random.default_rng(0).standard_normal((n, K))
    y = eng.apply(0, x); ref = oracle.apply_percell(mesh, 0, x)
    print('  apply rel', np.max(np.abs(y - ref)) / np.max(np.abs(ref)))
    x0 = inputs3[0, :n, :]
    b = eng.rhs(0, x0); bref = oracle.rhs_percell(mesh, 0, x0, inputs3[1])
    print('  rhs rel', np.max(np.abs(b - bref)) / np.max(np.abs(bref)))
    eng.set_state(x0)
    r = eng.step(0)
    xs = eng.get_state()[:n]
    res = bref - oracle.apply_percell(mesh, 0, xs)
    print('  solve: sweeps', r.sweeps, 'true residual (oracle operator)', np.max(np.linalg.norm(res, axis=0) / np.linalg.norm(bref, axis=0)))
    fx = os.path.join(root, 'tests', 'golden', 'config5_4m_expected.npz')
    if scale == 2 and os.path.exists(fx):
        e = np.load(fx)
        for ci in range(2):
            d = np.abs(xs[e['cells'], ci] - e['state'][0, ci])
            print('  vs fixture col', ci, 'max rel', d.max() / np.abs(e['state'][0, ci]).max())
    eng.close()
