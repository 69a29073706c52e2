"""CPU experiment for VERDICT r01 item 3: Chebyshev (three-term, dot-free) acceleration of the J^2 pass and of the
block-asynchronous pass.  x_{k+1} = w_{k+1} (g (G x_k + h - x_k) + x_k - x_{k-1}) + x_{k-1} with the spectrum of G assumed in
[a, b]; b from the measured contraction, a = 0 (J >= 0 entrywise; J^2's real spectrum is non-negative when J's is real).
Each accelerated pass reads one more vector (x_{k-1}): +128 of 483 B per row at K = 16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import hilbert_order, renumber_mesh
from oracle import cwr_oracle as orc

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
TR = 64
mesh = cw.synthetic.make_mesh(nx, nx, 3, seed=4, dt=dt, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
n = mesh['nreal'] + 1
order = hilbert_order(mesh['face_x'], mesh['face_y'], n)
mesh = renumber_mesh(mesh, order)
orc.derive_coefficients(mesh)
inp = cw.synthetic.distinct_input_array(mesh, 4, seed=4)
lhs = orc.LHS(mesh); lhs.update_values(mesh, 0)
A = lhs.csr().tocsr()[:n, :n]
D = A.diagonal()
J = sp.identity(n, format='csr') - sp.diags(1.0 / D) @ A
J.eliminate_zeros()
J2 = (J @ J).tocsr()
tile = np.arange(n) // TR
coo = J2.tocoo()
inside = tile[coo.row] == tile[coo.col]
Jin = sp.csr_matrix((coo.data[inside], (coo.row[inside], coo.col[inside])), shape=(n, n))
Jout = sp.csr_matrix((coo.data[~inside], (coo.row[~inside], coo.col[~inside])), shape=(n, n))
print('eigs of J (largest |.|):', end=' ')
try:
    from scipy.sparse.linalg import eigs
    ev = eigs(J, k=6, which='LM', return_eigenvectors=False, maxiter=5000, tol=1e-6)
    print(np.round(ev, 4))
except Exception as exc:
    print('eigs failed', exc)

for k in range(inp.shape[2]):
    x0 = inp[0, :n, k]
    r = orc.RHS(mesh, inp[:, :, k].copy()); r.update_values(x0, mesh, 0)
    bh = r.vals / D
    c2 = bh + J @ bh
    nb = np.linalg.norm(bh)
    def resid(x): return np.linalg.norm(bh - (x - J @ x)) / nb
    def plain(x): return c2 + J2 @ x
    def basync(x, L=2):
        g = c2 + Jout @ x; y = x
        for _ in range(L): y = g + Jin @ y
        return y
    for name, G in (('plain J^2', plain), ('block-async x2', basync)):
        x = x0.copy(); hist = []
        for p in range(1, 400):
            x = G(x); hist.append(resid(x))
            if hist[-1] < 1e-12: break
        rate = (hist[-1] / hist[max(0, len(hist) - 6)]) ** (1.0 / min(5, len(hist) - 1))
        out = [f'const {k} {name}: {p} passes (rate {rate:.3f})']
        for b in (rate, 0.8 * rate, 0.6 * rate):
            a = 0.0
            gam = 2.0 / (2.0 - b - a); sig = (b - a) / (2.0 - b - a)
            xm, x = x0.copy(), gam * (G(x0) - x0) + x0
            w = 1.0 / (1.0 - 0.5 * sig * sig)
            for q in range(2, 400):
                xn = w * (gam * (G(x) - x) + x - xm) + xm
                xm, x = x, xn
                w = 1.0 / (1.0 - 0.25 * sig * sig * w)
                rr = resid(x)
                if rr < 1e-12 or not np.isfinite(rr) or rr > 1e6: break
            out.append(f'cheb b={b:.3f}: {q if rr < 1e-12 else "fail"}')
        print('; '.join(out), flush=True)
