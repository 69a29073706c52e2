import sys, numpy as np, scipy.sparse as sp
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clearwater_riverine_amd as cw
from clearwater_riverine_amd.ordering import hilbert_order, renumber_mesh
from clearwater_riverine_amd.partition import partition_mesh
from clearwater_riverine_amd.distributed import auto_halo_depth
mesh = cw.synthetic.make_mesh(1000, 1000, 2, seed=4, dt=40.0, diffusion_coefficient=0.5)
n = mesh['nreal'] + 1
mesh = renumber_mesh(mesh, hilbert_order(mesh['face_x'], mesh['face_y'], n))
for world in (8, 4, 2):
    depth = auto_halo_depth(n, world)
    worst = 0; tot_rows = 0
    for r in range(world):
        lm = partition_mesh(mesh['edges_face1'], mesh['edges_face2'], n, world, r, depth=depth)
        nreal = lm.n_rows + lm.n_halo
        f1, f2 = lm.face1, lm.face2
        real = f2 < nreal
        a, b = f1[real], f2[real]
        A = sp.coo_matrix((np.ones(2 * len(a)), (np.r_[a, b], np.r_[b, a])), shape=(nreal, nreal)).tocsr()
        # rows with a J^2 row: all neighbours computed
        comp = np.arange(nreal) < lm.n_rows
        ok = np.asarray((A @ (~comp).astype(float))).ravel() == 0
        nsq = int(np.argmin(ok[: lm.n_rows])) if not ok[: lm.n_rows].all() else lm.n_rows
        A2 = (A[:nsq] @ A).tocsr()
        mx = 0
        for c0 in range(0, nsq, 64):
            c1 = min(c0 + 64, nsq)
            cols = np.unique(np.r_[A2.indices[A2.indptr[c0]:A2.indptr[c1]], np.arange(c0, c1)])
            mx = max(mx, len(cols))
        worst = max(worst, mx); tot_rows += nsq
        print(f'world {world} depth {depth} rank {r}: core {lm.n_core}, J^2 rows {nsq} (+{100*(nsq-lm.n_core)/lm.n_core:.1f} %), peers {len(lm.peers)}, max distinct rows per 64-row tile {mx}', flush=True)
    print(f'world {world}: worst tile {worst}, replayed rows overall +{100*(tot_rows-n)/n:.1f} %')
