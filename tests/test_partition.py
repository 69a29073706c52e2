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


@pytest.mark.parametrize('world', [1, 2, 3, 4, 8])
def test_partition_invariants_and_local_operator(world):
    from clearwater_riverine_amd.partition import partition_mesh, range_bounds
    mesh, inputs3 = make_case()
    f1, f2 = mesh['edges_face1'], mesh['edges_face2']
    n = mesh['nreal'] + 1
    parts = [partition_mesh(f1, f2, n, world, r) for r in range(world)]
    bounds = range_bounds(n, world)
    assert bounds[0] == 0 and bounds[-1] == n
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, 2))
    t = 1
    y_global = oracle.apply_percell(mesh, t, x)
    b_global = oracle.rhs_percell(mesh, t, x, inputs3[t + 1])
    owners_of_face1 = np.zeros(len(f1), dtype=int)
    for r, lm in enumerate(parts):
        assert (lm.lo, lm.hi) == (bounds[r], bounds[r + 1]) and lm.n_owned == lm.hi - lm.lo
        # numbering: owned | halo (sorted by global id == by owner) | ghost
        assert np.array_equal(lm.cell_global[:lm.n_owned], np.arange(lm.lo, lm.hi))
        halo = lm.cell_global[lm.n_owned:lm.n_real]
        assert np.all(np.diff(halo) > 0) and np.all((halo < lm.lo) | (halo >= lm.hi)) and np.all(halo < n)
        assert np.all(lm.cell_global[lm.n_real:] >= n)
        # faces: exactly those touching an owned cell, ascending global id, ids mapped back bit-exactly
        touching = ((f1 >= lm.lo) & (f1 < lm.hi)) | ((f2 >= lm.lo) & (f2 < lm.hi))
        assert np.array_equal(lm.edge_global, np.nonzero(touching)[0])
        assert np.array_equal(lm.cell_global[lm.face1], f1[lm.edge_global])
        assert np.array_equal(lm.cell_global[lm.face2], f2[lm.edge_global])
        assert lm.face1.max() < lm.n_real                                  # face1 real in local numbering
        owners_of_face1[(f1 >= lm.lo) & (f1 < lm.hi)] += 1
        # receive lists tile the halo block; peers never include self
        assert lm.recv_ptr[0] == 0 and lm.recv_ptr[-1] == lm.n_halo and r not in lm.peers
        # local operator and right-hand side rows == the global ones
        lmesh = local_oracle_mesh(mesh, lm)
        xl = x[lm.cell_global[:lm.n_real]]
        assert np.allclose(oracle.apply_percell(lmesh, t, xl)[:lm.n_owned], y_global[lm.lo:lm.hi], rtol=1e-13, atol=1e-13)
        gl = inputs3[t + 1][lm.cell_global]
        bl = oracle.rhs_percell(lmesh, t, xl, gl)
        assert np.allclose(bl[:lm.n_owned], b_global[lm.lo:lm.hi], rtol=1e-13, atol=1e-13)
    assert np.all(owners_of_face1 == 1)                                    # every face has exactly one face1 owner
    # send lists mirror receive lists
    for r, lm in enumerate(parts):
        for i, s in enumerate(lm.peers):
            sent = lm.cell_global[lm.send_cells[lm.send_ptr[i]:lm.send_ptr[i + 1]]]
            other = parts[s]
            j = list(other.peers).index(r)
            got = other.cell_global[other.n_owned + other.recv_ptr[j]: other.n_owned + other.recv_ptr[j + 1]]
            assert np.array_equal(sent, got)


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
        lm = partition_mesh(f1, f2, n, world, rank)
        x_global = np.random.default_rng(5).standard_normal((n, K))       # same on every rank (the check)
        vec = np.full((lm.n_real, K), np.nan)
        vec[:lm.n_owned] = x_global[lm.lo:lm.hi]                           # a rank knows only its own rows
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
            vec[lm.n_owned + lm.recv_ptr[i]: lm.n_owned + lm.recv_ptr[i + 1]] = rb.numpy()
        assert np.array_equal(vec, x_global[lm.cell_global[:lm.n_real]])   # halo rows bit-exact
        # operator rows of this rank, gathered over ranks == the global product
        from test_partition import local_oracle_mesh
        y_local = oracle.apply_percell(local_oracle_mesh(mesh, lm), 2, vec)[:lm.n_owned]
        parts = [None] * world
        dist.all_gather_object(parts, y_local)
        y = np.concatenate(parts, axis=0)
        want = oracle.apply_percell(mesh, 2, x_global)
        assert np.allclose(y, want, rtol=1e-13, atol=1e-13)
        # inner products complete with an all-reduce of per-rank partial sums
        part = torch.tensor((y_local * y_local).sum(axis=0))
        dist.all_reduce(part)
        assert np.allclose(part.numpy(), (want * want).sum(axis=0), rtol=1e-12)
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
