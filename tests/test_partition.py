"""CPU tests of the N > 1 path: the contiguous-range partition, its halo lists, and the exchange pattern
(world_size 2 over gloo), with the oracle standing in for the per-rank compute."""
import os
import socket

import numpy as np
import pytest

import cwr_oracle as oracle


def make_case(seed=3, K=2):
    import clearwater_riverine_amd as cw
    mesh = cw.synthetic.make_mesh(26, 9, 4, seed=seed, n_merge=20, shuffle_window=16)
    oracle.derive_coefficients(mesh)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K)
    return mesh, inputs3


def local_oracle_mesh(mesh, lm):
    """The oracle's mesh dict of one rank's local mesh (local numbering)."""
    eg, cg = lm.edge_global, lm.cell_global
    return {
        'edges_face1': lm.face1, 'edges_face2': lm.face2, 'nreal': lm.n_real - 1,
        'face_x': mesh['face_x'][cg], 'face_y': mesh['face_y'][cg],
        'advection_coeff': mesh['advection_coeff'][:, eg], 'coeff_to_diffusion': mesh['coeff_to_diffusion'][:, eg],
        'edge_velocity': mesh['edge_velocity'][:, eg], 'volume': mesh['volume'][:, cg], 'dt': mesh['dt'],
        'diffusion_coefficient': mesh['diffusion_coefficient'],
    }


@pytest.mark.parametrize('depth', [1, 2, 4, 12])
@pytest.mark.parametrize('world', [1, 2, 3, 4, 8])
def test_partition_invariants_and_local_operator(world, depth):
    from clearwater_riverine_amd.partition import partition_mesh, range_bounds
    mesh, inputs3 = make_case()
    f1, f2 = mesh['edges_face1'], mesh['edges_face2']
    n = mesh['nreal'] + 1
    parts = [partition_mesh(f1, f2, n, world, r, depth=depth) for r in range(world)]
    bounds = range_bounds(n, world)
    assert bounds[0] == 0 and bounds[-1] == n
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, 2))
    t = 1
    y_global = oracle.apply_percell(mesh, t, x)
    b_global = oracle.rhs_percell(mesh, t, x, inputs3[t + 1])
    owners_of_face1 = np.zeros(len(f1), dtype=int)
    for r, lm in enumerate(parts):
        assert (lm.lo, lm.hi) == (bounds[r], bounds[r + 1]) and lm.n_core == lm.hi - lm.lo
        assert lm.depth == (depth if world > 1 else 1)
        # numbering: core | computed halo layers | last (read-only) layer | ghost
        assert np.array_equal(lm.cell_global[:lm.n_core], np.arange(lm.lo, lm.hi))
        halo = lm.cell_global[lm.n_core:lm.n_real]
        assert len(np.unique(halo)) == len(halo) and np.all((halo < lm.lo) | (halo >= lm.hi)) and np.all(halo < n)
        assert np.all(lm.cell_global[lm.n_real:] >= n)
        if world == 1:
            assert lm.n_rows == lm.n_core and lm.n_halo == 0
        # faces: exactly those touching a computed row, ascending global id, ids mapped back bit-exactly
        comp = np.zeros(len(mesh['face_x']), dtype=bool)
        comp[lm.cell_global[:lm.n_rows]] = True
        touching = comp[f1] | comp[f2]
        assert np.array_equal(lm.edge_global, np.nonzero(touching)[0])
        assert np.array_equal(lm.cell_global[lm.face1], f1[lm.edge_global])
        assert np.array_equal(lm.cell_global[lm.face2], f2[lm.edge_global])
        assert lm.face1.max() < lm.n_real                                  # face1 real in local numbering
        owners_of_face1[(f1 >= lm.lo) & (f1 < lm.hi)] += 1
        # receive lists cover every halo row exactly once; peers never include self
        assert lm.recv_ptr[0] == 0 and lm.recv_ptr[-1] == lm.n_real - lm.n_core and r not in lm.peers
        assert sorted(lm.recv_cells) == list(range(lm.n_core, lm.n_real))
        # EVERY computed row (core and replayed halo layers) reproduces the global operator / right-hand side row
        lmesh = local_oracle_mesh(mesh, lm)
        xl = x[lm.cell_global[:lm.n_real]]
        rows_g = lm.cell_global[:lm.n_rows]
        assert np.allclose(oracle.apply_percell(lmesh, t, xl)[:lm.n_rows], y_global[rows_g], rtol=1e-13, atol=1e-13)
        gl = inputs3[t + 1][lm.cell_global]
        bl = oracle.rhs_percell(lmesh, t, xl, gl)
        assert np.allclose(bl[:lm.n_rows], b_global[rows_g], rtol=1e-13, atol=1e-13)
    assert np.all(owners_of_face1 == 1)                                    # every face has exactly one face1 owner
    # send lists mirror receive lists, element for element
    for r, lm in enumerate(parts):
        for i, s in enumerate(lm.peers):
            sent = lm.cell_global[lm.send_cells[lm.send_ptr[i]:lm.send_ptr[i + 1]]]
            other = parts[s]
            j = list(other.peers).index(r)
            got = other.cell_global[other.recv_cells[other.recv_ptr[j]: other.recv_ptr[j + 1]]]
            assert np.array_equal(sent, got)
            assert np.all((sent >= lm.lo) & (sent < lm.hi))


def test_deep_halo_sweeps_reproduce_global_jacobi():
    """s sweeps between exchanges: the core rows after every sweep equal the global Jacobi iterate and never read
    a stale value (numpy stand-in for the kernel; the validity-shrinks-one-layer-per-sweep argument, executed
    with NaN marking every stale row)."""
    import scipy.sparse as sp
    from clearwater_riverine_amd.partition import partition_mesh
    mesh, inputs3 = make_case(K=1)
    f1, f2 = mesh['edges_face1'], mesh['edges_face2']
    n = mesh['nreal'] + 1
    t, world, depth = 1, 3, 3
    lhs = oracle.LHS(mesh)
    lhs.update_values(mesh, t)
    A = lhs.csr()
    d = A.diagonal()
    J = sp.eye(n) - sp.diags(1.0 / d) @ A
    bh = oracle.rhs_percell(mesh, t, inputs3[t, :n, 0] + 1.0, inputs3[t + 1, :, 0]) / d
    xg = np.ones(n)
    parts = [partition_mesh(f1, f2, n, world, r, depth=depth) for r in range(world)]
    loc = []
    for lm in parts:
        rows = lm.cell_global[:lm.n_rows]
        Jl = J[rows][:, lm.cell_global[:lm.n_real]]
        loc.append({'lm': lm, 'J': sp.csr_matrix(Jl), 'b': bh[rows], 'x': np.full(lm.n_real, np.nan)})
    for L in loc:
        L['x'][:L['lm'].n_core] = xg[L['lm'].lo:L['lm'].hi]
    since = depth                                                          # forces the first exchange
    for sweep in range(9):
        if since >= depth:
            for L in loc:                                                  # exchange: owners' core rows -> all halo layers
                lm = L['lm']
                L['x'][lm.n_core:] = np.nan
            snapshot = np.concatenate([L['x'][:L['lm'].n_core] for L in loc])
            for L in loc:
                lm = L['lm']
                L['x'][lm.recv_cells] = snapshot[lm.cell_global[lm.recv_cells]]
            since = 0
        xg = bh + J @ xg
        for L in loc:
            lm = L['lm']
            new = np.full(lm.n_real, np.nan)
            with np.errstate(invalid='ignore'):
                new[:lm.n_rows] = L['b'] + L['J'] @ np.where(np.isnan(L['x']), 0.0, L['x'])
                # a row is valid only if every input it read was valid
                bad = (abs(L['J']) @ np.isnan(L['x']).astype(float)) > 0
            new[:lm.n_rows][bad] = np.nan
            L['x'] = new
            core = new[:lm.n_core]                                         # core rows: never invalid, equal to the global iterate
            assert not np.isnan(core).any()                                # (to rounding: scipy sums the sliced rows in another order)
            assert np.allclose(core, xg[lm.lo:lm.hi], rtol=1e-13, atol=0)
        since += 1


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ok_flags):
    import torch
    import torch.distributed as dist
    from clearwater_riverine_amd.partition import partition_mesh
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    try:
        mesh, inputs3 = make_case()
        f1, f2 = mesh['edges_face1'], mesh['edges_face2']
        n = mesh['nreal'] + 1
        K = 2
        lm = partition_mesh(f1, f2, n, world, rank, depth=2)
        x_global = np.random.default_rng(5).standard_normal((n, K))       # same on every rank (the check)
        vec = np.full((lm.n_real, K), np.nan)
        vec[:lm.n_core] = x_global[lm.lo:lm.hi]                            # a rank knows only its own rows
        # the engine's exchange pattern (csrc/cwr_engine.hip exchange_halo): pack rows per peer, grouped send/recv
        reqs, recv_bufs = [], []
        for i, peer in enumerate(lm.peers):
            rows = lm.send_cells[lm.send_ptr[i]:lm.send_ptr[i + 1]]
            sb = torch.from_numpy(np.ascontiguousarray(vec[rows]))
            rb = torch.empty((int(lm.recv_ptr[i + 1] - lm.recv_ptr[i]), K), dtype=torch.float64)
            recv_bufs.append(rb)
            reqs.append(dist.isend(sb, int(peer)))
            reqs.append(dist.irecv(rb, int(peer)))
        for q in reqs:
            q.wait()
        for i, rb in enumerate(recv_bufs):
            vec[lm.recv_cells[lm.recv_ptr[i]: lm.recv_ptr[i + 1]]] = rb.numpy()     # the engine's k_unpack_rows
        assert np.array_equal(vec, x_global[lm.cell_global[:lm.n_real]])   # halo rows bit-exact
        # operator rows of this rank, gathered over ranks == the global product
        from test_partition import local_oracle_mesh
        y_local = oracle.apply_percell(local_oracle_mesh(mesh, lm), 2, vec)[:lm.n_core]
        parts = [None] * world
        dist.all_gather_object(parts, y_local)
        y = np.concatenate(parts, axis=0)
        want = oracle.apply_percell(mesh, 2, x_global)
        assert np.allclose(y, want, rtol=1e-13, atol=1e-13)
        # inner products complete with an all-reduce of per-rank partial sums
        part = torch.tensor((y_local * y_local).sum(axis=0))
        dist.all_reduce(part)
        assert np.allclose(part.numpy(), (want * want).sum(axis=0), rtol=1e-12)
        # set-up side of a partitioned run (distributed.py): the curve order comes from rank 0 over the control plane, and
        # a rank's slices are cut straight from the reference arrays through composed index maps -- equal to slicing a
        # renumbered copy of the global mesh, which no rank builds any more
        from clearwater_riverine_amd.distributed import shared_hilbert_order
        from clearwater_riverine_amd.ordering import hilbert_order, renumber_mesh
        from clearwater_riverine_amd.partition import slice_fields
        from clearwater_riverine_amd.ordering import balance_windows
        from clearwater_riverine_amd.engine import tile_rows
        order = shared_hilbert_order(mesh, n, rank, world, K=16)
        assert tile_rows(16) == 64 and tile_rows(1) == 256 and tile_rows(8) == 128 and tile_rows(2) == 256
        assert np.array_equal(order, balance_windows(hilbert_order(mesh['face_x'], mesh['face_y'], n), f1, f2, window=64))
        assert np.array_equal(np.sort(order), np.arange(n))
        rm = renumber_mesh(mesh, order)
        lm2 = partition_mesh(rm['edges_face1'], rm['edges_face2'], n, world, rank, depth=3)
        want_f = slice_fields(lm2, rm, np.zeros(len(f1)))
        ref_cells = np.where(lm2.cell_global < n, order[np.minimum(lm2.cell_global, n - 1)], lm2.cell_global)
        got_f = slice_fields(lm2, mesh, np.zeros(len(f1)), ref_cells)
        assert all(np.array_equal(got_f[k], want_f[k]) for k in want_f)
        ok_flags[rank] = 1
    finally:
        dist.destroy_process_group()


def test_halo_exchange_world_size_2_gloo():
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    ok = ctx.Array('i', [0] * world)
    procs = [ctx.Process(target=_worker, args=(r, world, port, ok)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    for p in procs:
        if p.is_alive():
            p.terminate()
    assert list(ok) == [1] * world


def test_auto_halo_depth_tracks_the_per_rank_size():
    from clearwater_riverine_amd.distributed import auto_halo_depth
    assert auto_halo_depth(1_000_000, 1) == 1
    assert [auto_halo_depth(1_000_000, w) for w in (2, 4, 8)] == [16, 16, 14]
    assert auto_halo_depth(10_000, 4) == 8 and auto_halo_depth(64_000_000, 8) == 16
    assert all(auto_halo_depth(n, w) % 2 == 0 for n in (5_000, 123_456, 9_999_999) for w in (2, 3, 8))


def test_group_layout_and_constituent_slices():
    """distributed.group_layout / ConstituentSlice (round 6: N GPUs as cell ranges x constituent groups): every (group, range) pair occurs once,
    the groups' column ranges tile [0, K) in order, ranges vary fastest; a slice of an input provider is the columns of the dense array."""
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.distributed import ConstituentSlice, group_layout
    for world, G, K in [(8, 2, 16), (8, 4, 16), (8, 8, 16), (4, 2, 5), (6, 3, 7), (8, 1, 3)]:
        seen, cols = set(), {}
        for rank in range(world):
            g, r, R, k0, k1 = group_layout(rank, world, G, K)
            assert R == world // G and (g, r) == divmod(rank, R) and 0 <= k0 < k1 <= K
            seen.add((g, r)); cols.setdefault(g, (k0, k1))
            assert cols[g] == (k0, k1)
        assert len(seen) == world
        cuts = [cols[g] for g in range(G)]
        assert cuts[0][0] == 0 and cuts[-1][1] == K and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1
    for bad in [(8, 3, 16), (4, 8, 16), (4, 4, 3)]:
        with pytest.raises(ValueError):
            group_layout(0, *bad)
    mesh = cw.synthetic.make_mesh(12, 8, 3, seed=2)
    prov = cw.synthetic.DistinctInputs(mesh, 6, seed=1)
    dense = prov.dense()
    n = mesh['nreal'] + 1
    sl = ConstituentSlice(prov, 2, 5)
    assert sl.shape == dense.shape[:2] + (3,)
    cells = np.array([0, 5, n - 1])
    assert np.array_equal(sl.initial_rows(cells), dense[0, cells, 2:5])
    ghosts = np.arange(n, dense.shape[1])
    assert np.array_equal(sl.ghost_columns(ghosts), dense[:, n:, 2:5])
    assert sl.real_input_entries(cells)[2].shape == (0, 3)
