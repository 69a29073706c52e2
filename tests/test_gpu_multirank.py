"""GPU test of the partitioned (N > 1) solver loop with several ranks on ONE GPU.

Real RCCL refuses two ranks on one device, and the GPU box has one GPU, so the engine is pointed
(CWR_RCCL_LIB) at tests/mock_rccl/libmock_rccl.so, a shared-memory stand-in for the nine RCCL entry
points it uses.  Everything else is the product path: partition.py, the C-ABI engine with halo rows,
k_pack_rows, the grouped send/recv before every operator launch, the all-reduces of the inner products,
and the convergence decisions taken identically on every rank.

Round 3: the stand-in is STREAM-ASYNCHRONOUS by default (mock_rccl.cpp: hipStreamWaitValue64 / hipMemcpyAsync /
hipStreamWriteValue64 on the caller's stream over a page-locked shared segment; no host-side stream drain), so the two-stream /
two-event plumbing of the overlapped exchange is exercised the way RCCL exercises it.  CWR_MOCK_ASYNC=0 restores the
host-synchronous form; CWR_MOCK_ASYNC=2 refuses to fall back.
"""
import multiprocessing as mp
import os
import queue
import subprocess

import numpy as np
import pytest

import cwr_oracle as oracle
from util import flux_err, rel_err

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
MOCK_SRC = os.path.join(HERE, 'mock_rccl', 'mock_rccl.cpp')
MOCK_LIB = os.path.join(HERE, 'mock_rccl', 'libmock_rccl.so')


def build_mock():
    if not os.path.exists(MOCK_LIB) or os.path.getmtime(MOCK_LIB) < os.path.getmtime(MOCK_SRC):
        subprocess.run(['/opt/rocm/bin/hipcc', '-O2', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', MOCK_SRC, '-o', MOCK_LIB, '-lrt'],
                       check=True)
    return MOCK_LIB


def make_case(K):
    import clearwater_riverine_amd as cw
    if os.environ.get('CWR_TEST_SOAK_SEED'):     # the partitioned soak: a random mesh from the seed (every rank and the parent build the same one)
        rng = np.random.default_rng(int(os.environ['CWR_TEST_SOAK_SEED']))
        nx, ny = int(rng.integers(60, 200)), int(rng.integers(30, 110))
        nb = nx * ny
        mesh = cw.synthetic.make_mesh(nx, ny, 4, seed=int(rng.integers(1, 10**6)), n_merge=int(rng.choice([0, nb // 40, nb // 20])),
                                      n_merge4=int(rng.choice([0, nb // 300])), n_dry=int(rng.choice([0, 2, nb // 100])), shuffle_window=int(rng.choice([16, 32])),
                                      dt=float(rng.choice([10.0, 40.0, 40.0, 400.0])), diffusion_coefficient=float(rng.choice([0.1, 0.5, 2.0])))
        return mesh, cw.synthetic.distinct_input_array(mesh, K, seed=int(rng.integers(1, 1000)))
    if os.environ.get('CWR_TEST_BIG'):           # several 64-row tiles per rank, stiff enough for ~50 sweeps
        mesh = cw.synthetic.make_mesh(160, 96, 4, seed=22, n_merge=200, shuffle_window=16, dt=40.0, diffusion_coefficient=0.5)
    else:
        mesh = cw.synthetic.make_mesh(48, 20, 4, seed=21, n_merge=60, shuffle_window=16, n_dry=2)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K)
    if os.environ.get('CWR_TEST_SOURCES'):       # point sources: non-zero input_array entries on REAL cells at levels >= 1
        n = mesh['nreal'] + 1
        rng = np.random.default_rng(11)
        src = rng.choice(n, size=9, replace=False)
        inputs3[1:3, src[:5], 0] = 250.0
        inputs3[2, src[4:], K - 1] = 40.0 + rng.random(5)
    return mesh, inputs3


def case_lines(mesh):
    """Three boundary-condition lines: the ghost faces of the mesh dealt round-robin (reference face ids)."""
    gf = np.nonzero(np.asarray(mesh['edges_face2']) > mesh['nreal'])[0]
    return [gf[0::3], gf[1::3], gf[2::3]]


def _rank_body(rank, world, K, solver, depth, uid):
    """Three steps of one rank of a partitioned run; returns the tuple the tests read (see run_ranks)."""
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.distributed import PartitionedTransport
    mesh, inputs3 = make_case(K)
    fw = int(os.environ.get('CWR_TEST_FLOW_WINDOW', '0'))     # > 0: every rank keeps a ring of this many levels of ITS slices (round 6)
    if fw and os.environ.get('CWR_TEST_LAZY'):                # ... fed from a LEVEL SOURCE over the whole mesh instead of the rank's resident slices
        full = {k: np.asarray(mesh[k]) for k in ('face_flow', 'edge_velocity', 'volume')}
        calls = []

        def source(t0, t1):
            calls.append((t0, t1))
            return full['face_flow'][t0:t1], full['edge_velocity'][t0:t1], full['volume'][t0:t1]
        mesh = {k: v for k, v in mesh.items() if k not in ('face_flow', 'edge_velocity', 'volume', 'advection_coeff', 'coeff_to_diffusion')}
        mesh['level_source'] = source
    pt = PartitionedTransport(mesh, inputs3, rank, world, device=0, unique_id=uid, halo_depth=depth,
                              renumber='hilbert' if depth >= 4 else None, flow_window=fw or None)
    infos, comm_counts, kinds = [], [], []
    pt.set_boundary_lines(case_lines(mesh))
    mass0 = pt.engine.domain_mass(0)
    det = bool(os.environ.get('CWR_TEST_DETERMINISTIC'))      # CWR_STEP_DETERMINISTIC on every step of every rank
    for t in range(3):
        r = pt.step(t, tol=1e-12, mass_flux=True, solver=solver, mass_balance=True, deterministic=det)
        infos.append((r.sweeps, r.iterations))
        comm_counts.append((r.exchanges, r.overlapped, r.checks))
        kinds.append(r.chained)
    adv, dif, tot = pt.engine.get_mass_flux()
    owned_faces = pt.local.face1 < pt.local.n_core
    overlapped = pt.engine.comm_selftest(257)             # self send/recv through the stand-in + the overlap counter
    import ctypes
    is_async = ctypes.CDLL(MOCK_LIB).mockRcclLastCommAsync()      # 1: the stand-in only enqueued on the engine's streams
    out = (rank, pt.owned_reference_ids(), pt.local.hi, pt.owned_state(), pt.local.edge_global[owned_faces],
           tot[owned_faces], infos, None, pt.engine.get_mass_balance(), mass0, pt.engine.domain_mass(3), overlapped,
           is_async, comm_counts, pt.engine.get_tile_schedule()[0] is not None, kinds)
    return pt, out


def _rank_main(rank, world, K, solver, depth, uid_pipe, out_queue):
    try:
        os.environ['CWR_RCCL_LIB'] = MOCK_LIB
        import clearwater_riverine_amd as cw
        if rank == 0:
            uid = cw.TransportEngine.comm_unique_id()
            for _ in range(world - 1):
                uid_pipe.put(uid)
        else:
            uid = uid_pipe.get(timeout=120)
        pt, out = _rank_body(rank, world, K, solver, depth, uid)
        out_queue.put(out)
        pt.close()
    except Exception as exc:                                  # surface the failure in the parent
        out_queue.put((rank, 0, 0, None, None, None, None, repr(exc), None, None, None, 0, -1, None, False, None))


def _host_main(host, world, per_host, K, solver, depth, uid_pipe, out_queue):
    """One PROCESS hosting `per_host` consecutive ranks as threads (ranks host * per_host ...): how 8 ranks run on the one-GPU
    box, which allows at most 6 processes on its card.  Every rank is a complete engine with its own streams and communicator;
    the C-ABI calls release the GIL, so the ranks of a process block in their collectives independently."""
    import threading
    ranks = list(range(host * per_host, min(world, (host + 1) * per_host)))
    try:
        os.environ['CWR_RCCL_LIB'] = MOCK_LIB
        import clearwater_riverine_amd as cw
        uid = cw.TransportEngine.comm_unique_id()             # (every host calls it once: loads the stand-in before threads start)
        n_hosts = -(-world // per_host)
        if host == 0:
            for _ in range(n_hosts - 1):
                uid_pipe.put(uid)
        else:
            uid = uid_pipe.get(timeout=120)
    except Exception as exc:
        for r in ranks:
            out_queue.put((r, 0, 0, None, None, None, None, repr(exc), None, None, None, 0, -1, None, False, None))
        return
    engines, lock = [], threading.Lock()

    def body(rank):
        try:
            pt, out = _rank_body(rank, world, K, solver, depth, uid)
            with lock:
                engines.append(pt)
            out_queue.put(out)
        except Exception as exc:
            out_queue.put((rank, 0, 0, None, None, None, None, repr(exc), None, None, None, 0, -1, None, False, None))
    threads = [threading.Thread(target=body, args=(r,)) for r in ranks]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for pt in engines:                                        # (every rank of this host is done stepping: the communicators go together)
        pt.engine.close()


def run_ranks(world, target, args, per_host=1):
    """Spawn `world` rank processes, collect one result per rank, and ALWAYS reap the children: a rank that dies without
    posting a result (segfault, mock-RCCL timeout) must not leave its peers alive on the GPU box.
    per_host > 1: ceil(world / per_host) processes, each hosting per_host ranks as threads (target = _host_main)."""
    ctx = mp.get_context('spawn')
    uid_pipe, out_queue = ctx.Queue(), ctx.Queue()
    if per_host > 1:
        procs = [ctx.Process(target=target, args=(h, world, per_host) + tuple(args) + (uid_pipe, out_queue)) for h in range(-(-world // per_host))]
    else:
        procs = [ctx.Process(target=target, args=(r, world) + tuple(args) + (uid_pipe, out_queue)) for r in range(world)]
    results = []
    try:
        for p in procs:
            p.start()
        import time
        deadline = time.monotonic() + 240
        while len(results) < world and time.monotonic() < deadline:
            try:
                results.append(out_queue.get(timeout=1))
            except queue.Empty:
                # a rank that died without posting (segfault, abort) will never post: stop waiting for it
                if any(p.exitcode not in (None, 0) for p in procs):
                    break
    finally:
        for p in procs:
            p.join(5 if len(results) < world else 60)
            if p.is_alive():
                p.terminate()
                p.join(10)
            if p.is_alive():
                p.kill()
                p.join(10)
    assert len(results) == world, f'{world - len(results)} rank(s) posted no result; exit codes {[p.exitcode for p in procs]}'
    errs = [r[7] for r in results if r[7]]
    assert not errs, errs
    results.sort(key=lambda r: r[0])
    return results


@pytest.mark.parametrize('world,K,solver,depth', [(2, 3, 'jacobi', 1), (2, 3, 'bicgstab', 1), (3, 16, 'auto', 1), (4, 1, 'auto', 1),
                                                  (2, 3, 'jacobi', 4), (3, 2, 'bicgstab', 3), (4, 16, 'auto', 8), (3, 1, 'auto', 2),
                                                  (2, 16, 'jacobi', 8), (3, 8, 'jacobi', 2), (2, 12, 'jacobi', 5)])
@pytest.mark.parametrize('local_reps', ['1', 'default'])
def test_partitioned_step_matches_single_rank_and_oracle(gpu_lib, world, K, solver, depth, local_reps, monkeypatch):
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')      # the single-rank reference run takes the same multi-launch path as the ranks
    monkeypatch.setenv('CWR_TWO_CLOSING', '1')   # ... and the ranks' batch shape (even passes + two closing sweeps): bitwise comparison
    if local_reps == '1':
        monkeypatch.setenv('CWR_LOCAL_REPS', '1')   # exact Jacobi passes (spawned ranks inherit the environment)
    elif solver == 'bicgstab' or depth < 2:
        pytest.skip('block-asynchronous passes only exist in J^2 sweeps (halo depth >= 2)')
    results = run_ranks(world, _rank_main, (K, solver, depth))
    mesh, inputs3 = make_case(K)
    n = mesh['nreal'] + 1
    state = np.full((n, K), np.nan)
    for r in results:
        state[r[1]] = r[3]                                   # rows by reference cell id (ranks may work in a Hilbert numbering)
    assert not np.isnan(state).any() and sum(len(r[1]) for r in results) == n
    # every rank took the same solver decisions (sweep / iteration counts)
    assert all(r[6] == results[0][6] for r in results)
    # single-rank HIP result
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.distributed import PartitionedTransport
    single = PartitionedTransport(mesh, inputs3, 0, 1)
    single.set_boundary_lines(case_lines(mesh))
    single_mass0 = single.domain_mass(0)
    for t in range(3):
        single.step(t, tol=1e-12, mass_flux=True, solver=solver, mass_balance=True)
    # output side (8f-4): every rank adds the boundary faces / cells it owns; the host adds the ranks
    ledger = sum(r[8] for r in results)
    assert np.allclose(ledger, single.mass_balance(), rtol=1e-9, atol=1e-12, equal_nan=True)
    for idx, ref_val in ((9, single_mass0), (10, single.domain_mass(3))):
        assert np.allclose(sum(r[idx][0] for r in results), ref_val[0], rtol=1e-9)
        assert sum(r[idx][1] for r in results) == pytest.approx(ref_val[1], rel=1e-12)
    single_state = single.gather_state()                     # reference numbering whatever the internal one
    assert rel_err(state, single_state) <= 1e-10
    if solver == 'jacobi' and depth >= 2 and local_reps == '1':   # same kernels on both sides (depth 1 cannot host J^2 passes):
        # exact Jacobi sweeps replay the owner's arithmetic, independent of world size, halo depth and numbering.
        # (The default block-asynchronous passes depend on the tiling: equal to solver tolerance, checked above.)
        assert np.array_equal(state, single_state)
    # oracle
    oracle.derive_coefficients(mesh)
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for _ in range(3):
        ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[3, :n] for k in range(K)], axis=1)
    assert rel_err(state, want) <= 1e-9
    # mass flux of the last step: each face reported by the owner of its face1
    tot = np.full((len(mesh['edges_face1']), K), np.nan)
    for r in results:
        tot[r[4]] = r[5]
    want_flux = np.stack([ref.constituent_dict[f'c{k}'].total_mass_flux[2] for k in range(K)], axis=1)
    assert flux_err(tot, want_flux) <= 1e-8


@pytest.mark.parametrize('K,depth', [(16, 8), (1, 6)])
def test_eight_ranks_through_the_stand_in(gpu_lib, K, depth, monkeypatch):
    """the target machine is one 8-GPU node, and 8 ranks had never executed by any route.  The one-GPU box
    allows 6 processes on its card, so the 8 ranks are 4 processes x 2 rank threads (_host_main): eight complete engines, eight
    communicators, the real partition of 8 with its middle ranks (up to six peers, two cut sides, few or no interior tiles) and its
    end ranks.  Oracle parity, every rank the same solver decisions and the same exchange / overlap / check counts per step.
    Two ranks in ONE process is an arrangement of this test only (the product runs one engine per process), and it limits the test
    in one way: the streams of one process share its in-order copy-engine rings, so the stream-asynchronous stand-in, which parks a
    copy behind a wait on a peer, lets one rank's receive block its process mate's send (gpurun_out/r04e / r04g)
    -> the stand-in runs in its HOST-synchronous mode here.  hipGraphs are ON again since round 5 (K = 16): while one rank thread
    captured, the runtime refused the other thread's calls ("operation not permitted when stream is capturing", r04f) as long as
    that thread's capture-interaction mode was the default; every ABI entry now puts its thread into the thread-local mode.
    Up to 6 ranks (one process each) the stream-asynchronous mode is what every other test uses, the poison tests included."""
    build_mock()
    world = 8
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_TEST_BIG', '1')
    if K == 1:
        monkeypatch.setenv('CWR_NO_GRAPHS', '1')         # (one of the two cases keeps round 4's arrangement)
    else:
        monkeypatch.delenv('CWR_NO_GRAPHS', raising=False)
    monkeypatch.setenv('CWR_MOCK_ASYNC', '0')
    monkeypatch.setenv('CWR_MOCK_TIMEOUT_S', '45')
    results = run_ranks(world, _host_main, (K, 'jacobi', depth), per_host=2)
    assert [r[0] for r in results] == list(range(world))
    assert all(r[12] == 0 for r in results)
    mesh, inputs3 = make_case(K)
    n = mesh['nreal'] + 1
    state = np.full((n, K), np.nan)
    tot = np.full((len(mesh['edges_face1']), K), np.nan)
    for r in results:
        state[r[1]] = r[3]
        tot[r[4]] = r[5]
    assert not np.isnan(state).any() and sum(len(r[1]) for r in results) == n
    assert all(r[6] == results[0][6] for r in results), [r[6] for r in results]        # sweeps / iterations per step
    # exchanges and checks per step are the same calls on every rank; how many exchanges ran BESIDE interior tiles is each rank's own
    # (a middle rank of eight may have no interior tile at all)
    assert all([(e_, c_) for e_, _, c_ in r[13]] == [(e_, c_) for e_, _, c_ in results[0][13]] for r in results), [r[13] for r in results]
    for r in results:
        for (sweeps, its), (exch, over, checks) in list(zip(r[6], r[13]))[1:]:
            passes = (sweeps - 1) // 2
            assert its == 0 and sweeps >= 20, (sweeps, its)                            # the case really iterates, on the sweep path
            assert 0 < exch <= -(-passes // max(1, depth // 2)) + 4, (sweeps, exch, depth)
            assert 0 <= over <= exch and 1 <= checks <= 2, (over, exch, checks)
    # some rank of the eight has interior tiles and ran its in-loop exchanges beside them
    assert any(r[11] > 0 for r in results), [r[11] for r in results]
    oracle.derive_coefficients(mesh)
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for _ in range(3):
        ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[3, :n] for k in range(K)], axis=1)
    assert rel_err(state, want) <= 1e-9
    want_flux = np.stack([ref.constituent_dict[f'c{k}'].total_mass_flux[2] for k in range(K)], axis=1)
    assert flux_err(tot, want_flux) <= 1e-8
    # the single-rank engine on the same mesh: the same answer to solver tolerance, and a comparable number of sweeps
    from clearwater_riverine_amd.distributed import PartitionedTransport
    single = PartitionedTransport(mesh, inputs3, 0, 1)
    single_sweeps = [single.step(t, tol=1e-12, mass_flux=True, solver='jacobi').sweeps for t in range(3)]
    assert rel_err(state, single.gather_state()) <= 1e-10
    got = [s_ for s_, _ in results[0][6]]
    assert all(g <= 1.5 * s_ + 4 for g, s_ in zip(got[1:], single_sweeps[1:])), (got, single_sweeps)


def _rank_config4(rank, world, uid):
    """One rank of BASELINE config 4 as it is specified: the 1 M-cell floodplain mesh, ONE tracer, contiguous id ranges with
    the automatic halo depth."""
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.distributed import PartitionedTransport
    exp = np.load(os.path.join(HERE, 'golden', 'config4_1m_expected.npz'))
    steps = int(exp['steps'])
    mesh = cw.synthetic.bench_mesh(steps + 1)
    inputs3 = np.ascontiguousarray(cw.synthetic.distinct_input_array(mesh, int(exp['K']), seed=cw.synthetic.BENCH_SEED)[:, :, :1])
    pt = PartitionedTransport(mesh, inputs3, rank, world, device=0, unique_id=uid, halo_depth=0, renumber='hilbert')
    owned_faces = pt.local.face1 < pt.local.n_core
    states, fluxes, infos = [], [], []
    for t in range(steps):
        r = pt.step(t, tol=1e-12, mass_flux=True)
        infos.append((r.sweeps, r.iterations, r.exchanges, r.overlapped, r.checks, r.flags, r.sweep_kernel))
        states.append(pt.owned_state()[:, 0].copy())
        fluxes.append(pt.engine.get_mass_flux()[2][owned_faces, 0].copy())
    out = (rank, pt.owned_reference_ids(), pt.local.depth, states, pt.local.edge_global[owned_faces], fluxes, infos, None,
           pt.local.n_core, pt.local.n_rows, len(pt.local.peers))
    return pt, out


def _rank_config5(rank, world, uid):
    """One rank of BASELINE config 5: the 4 M-cell mesh, 16 constituents, a K x K reaction applied on the device to the rank's own
    rows before the second step (test_gpu_fullsize.py: the same two levels the committed fixture holds)."""
    import importlib.util
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.distributed import PartitionedTransport
    spec = importlib.util.spec_from_file_location('make_expected_large', os.path.join(HERE, 'golden', 'make_expected_large.py'))
    large = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(large)
    exp = np.load(os.path.join(HERE, 'golden', 'config5_4m_expected.npz'))
    K, dt, steps = int(exp['K']), float(exp['dt']), int(exp['steps'])
    mesh = cw.synthetic.bench_mesh(steps, scale=2)
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=cw.synthetic.BENCH_SEED + 1)
    M = large.reaction_matrix(K, dt)
    cols = [int(k) for k in exp['cols']]
    pt = PartitionedTransport(mesh, inputs3, rank, world, device=0, unique_id=uid, halo_depth=0, renumber='hilbert')
    infos, states = [], []
    r = pt.step(0, tol=1e-12, mass_flux=True)
    infos.append((r.sweeps, r.iterations, r.exchanges, r.overlapped, r.checks, r.flags, r.sweep_kernel, r.chained))
    states.append(pt.owned_state()[:, cols].copy())
    pt.engine.react_linear(M)                                    # the device reaction hook, on this rank's own rows
    r = pt.step(1, tol=1e-12, mass_flux=True)
    infos.append((r.sweeps, r.iterations, r.exchanges, r.overlapped, r.checks, r.flags, r.sweep_kernel, r.chained))
    states.append(pt.owned_state()[:, cols].copy())
    out = (rank, pt.owned_reference_ids(), pt.local.depth, states, None, None, infos, None, pt.local.n_core, pt.local.n_rows, len(pt.local.peers),
           pt.numbering, np.diag(M)[cols])
    return pt, out


def _host_config4(host, world, per_host, uid_pipe, out_queue, which=4):
    import threading
    ranks = list(range(host * per_host, min(world, (host + 1) * per_host)))
    try:
        os.environ['CWR_RCCL_LIB'] = MOCK_LIB
        import clearwater_riverine_amd as cw
        uid = cw.TransportEngine.comm_unique_id()
        if host == 0:
            for _ in range(-(-world // per_host) - 1):
                uid_pipe.put(uid)
        else:
            uid = uid_pipe.get(timeout=240)
    except Exception as exc:
        for r in ranks:
            out_queue.put((r, None, 0, None, None, None, None, repr(exc)))
        return
    engines, lock = [], threading.Lock()

    def body(rank):
        try:
            pt, out = (_rank_config4 if which == 4 else _rank_config5)(rank, world, uid)
            with lock:
                engines.append(pt)
            out_queue.put(out)
        except Exception as exc:
            out_queue.put((rank, None, 0, None, None, None, None, repr(exc)))
    threads = [threading.Thread(target=body, args=(r,)) for r in ranks]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for pt in engines:
        pt.engine.close()


def test_config4_as_specified_one_tracer_on_the_1m_cell_mesh_across_eight_ranks(gpu_lib, monkeypatch):
    """BASELINE config 4 literally: "synthetic 1 M-cell unstructured floodplain mesh, 1 tracer, domain-decomposed across 8 x MI355X
    with ghost-halo RCCL" -- the decomposition, the halo exchanges and the solver loop of eight ranks on the FULL mesh (124 992-125 056
    cells per rank, halo depth 14, ping-pong passes with overlapped exchanges), element-wise against the committed SuperLU output of
    the full mesh (tests/golden/config4_1m_expected.npz: 65 536 sampled cells over the plume's decades, every ghost cell, whole-column
    norms, 16 384 sampled face fluxes).  Eight engines on ONE GPU through the stand-in (4 processes x 2 rank threads, see
    test_eight_ranks_through_the_stand_in): functional, not a measurement -- the one thing of config 4 that needs eight GPUs is its speed."""
    build_mock()
    world = 8
    monkeypatch.setenv('CWR_NO_GRAPHS', '1')
    monkeypatch.setenv('CWR_MOCK_ASYNC', '0')
    monkeypatch.setenv('CWR_MOCK_TIMEOUT_S', '90')
    results = _run_hosts(_host_config4, (4,), world, 2, 420)
    import clearwater_riverine_amd as cw
    exp = np.load(os.path.join(HERE, 'golden', 'config4_1m_expected.npz'))
    steps = int(exp['steps'])
    n, E = 1_000_000, None
    assert sum(len(r[1]) for r in results) == n and all(r[2] == 14 for r in results)       # every cell owned once; the automatic halo depth
    assert all(124_000 <= r[8] <= 126_000 for r in results) and all(r[10] >= 2 for r in results)
    assert all([i[:2] for i in r[6]] == [i[:2] for i in results[0][6]] for r in results)     # the same sweeps on every rank
    for r in results:
        for sweeps, its, exch, over, checks, flags, kern in r[6]:
            assert its == 0 and flags == 0 and kern == 6 and sweeps >= 20 and 0 < over <= exch and 1 <= checks <= 3, r[6]
    mesh_faces = 2_054_728
    for s_ in range(steps):
        col = np.full(n, np.nan)
        tot = np.full(mesh_faces, np.nan)
        seen = np.zeros(mesh_faces, bool)
        for r in results:
            col[r[1]] = r[3][s_]
            tot[r[4]] = r[5][s_]
            seen[r[4]] = True
        assert not np.isnan(col).any() and seen.all()            # (a boundary face without a boundary value has a NaN flux, as in the reference)
        assert rel_err(col[exp['cells']], exp['state'][s_, 0]) <= 1e-9                       # element-wise 1e-6 |b| + 1e-12 peak inside
        got = np.array([np.linalg.norm(col), np.sum(col), np.max(np.abs(col))])
        assert np.allclose(got, exp['norms'][s_, 0], rtol=1e-9, atol=0.0)                    # the WHOLE column
        assert flux_err(tot[exp['flux_faces']], exp['total_flux'][s_, 0]) <= 1e-8


def _run_hosts(target, args, world, per_host, timeout_s):
    ctx = mp.get_context('spawn')
    uid_pipe, out_queue = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=target, args=(h, world, per_host, uid_pipe, out_queue) + tuple(args)) for h in range(-(-world // per_host))]
    results = []
    try:
        for p in procs:
            p.start()
        import time
        deadline = time.monotonic() + timeout_s
        while len(results) < world and time.monotonic() < deadline:
            try:
                results.append(out_queue.get(timeout=1))
            except queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):
                    break
    finally:
        for p in procs:
            p.join(5 if len(results) < world else 180)
            if p.is_alive():
                p.terminate(); p.join(10)
            if p.is_alive():
                p.kill(); p.join(10)
    assert len(results) == world, f'{world - len(results)} rank(s) posted no result; exit codes {[p.exitcode for p in procs]}'
    errs = [r[7] for r in results if r[7]]
    assert not errs, errs
    results.sort(key=lambda r: r[0])
    return results


def test_config5_as_specified_4m_cells_16_constituents_device_reaction_across_eight_ranks(gpu_lib, monkeypatch):
    """BASELINE config 5: "synthetic 4 M-cell mesh, 16 constituents + per-step TSM / NSM reaction callback, 8 x MI355X" -- eight ranks of
    500 k cells each: large enough to CHAIN their tiles (lane-major numbering, interior and cut tiles chained separately, exchanges
    beside the interior lists), the reaction on the device between the steps (every rank on its own rows; the start-of-step exchange
    then carries the reacted halo rows).  Both levels of the committed SuperLU fixture (config5_4m_expected.npz: plain step, step
    behind the reaction; pulse and plume columns; 32 768 sampled cells element-wise + whole-column norms).  Functional: eight
    engines share one GPU through the stand-in (4 processes x 2 rank threads)."""
    build_mock()
    world = 8
    monkeypatch.setenv('CWR_NO_GRAPHS', '1')
    monkeypatch.setenv('CWR_MOCK_ASYNC', '0')
    monkeypatch.setenv('CWR_MOCK_TIMEOUT_S', '180')
    results = _run_hosts(_host_config4, (5,), world, 2, 900)
    exp = np.load(os.path.join(HERE, 'golden', 'config5_4m_expected.npz'))
    n = 4_000_000
    assert sum(len(r[1]) for r in results) == n and all(r[2] == 16 for r in results)
    assert all(r[11].startswith('lanes') for r in results), [r[11] for r in results]
    assert all([i[:2] for i in r[6]] == [i[:2] for i in results[0][6]] for r in results)
    for r in results:
        for sweeps, its, exch, over, checks, flags, kern, chained in r[6]:
            assert its == 0 and flags == 0 and kern == 6 and chained == 1 and 0 < over <= exch and 1 <= checks <= 4, r[6]   # (the first step has no sweep history: up to four batches since the ranks hold the row-wise error factor, round 5)
    diagM = results[0][12]
    for level in range(2):
        for ci in range(len(exp['cols'])):
            col = np.full(n, np.nan)
            for r in results:
                col[r[1]] = r[3][level][:, ci]
            assert not np.isnan(col).any()
            if level == 0:                                       # (the fixture's level 1 holds the override M[k, k] x the solved level: test_gpu_fullsize.py)
                col = col * diagM[ci]
            assert rel_err(col[exp['cells']], exp['state'][level, ci]) <= 1e-9          # element-wise 1e-6 |b| + 1e-12 peak inside
            got = np.array([np.linalg.norm(col), np.sum(col), np.max(np.abs(col))])
            assert np.allclose(got, exp['norms'][level, ci], rtol=1e-9, atol=0.0)


@pytest.mark.parametrize('world,K,depth', [(2, 4, 8), (4, 16, 8), (3, 1, 6), (2, 16, 14), (4, 4, 16)])
def test_partitioned_block_asynchronous_passes_keep_the_single_rank_sweep_count(gpu_lib, world, K, depth, monkeypatch):
    """Deep halos + tile-local re-application: the replayed layers (tiled like the core) and the never-computed outer
    layers of BOTH ping-pong vectors must hold this exchange's values, otherwise old iterates leak into the core and
    the solve needs many more passes (seen: 78 instead of 58 sweeps at 2 ranks, no convergence at 4)."""
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_TEST_BIG', '1')
    results = run_ranks(world, _rank_main, (K, 'jacobi', depth))
    from clearwater_riverine_amd.distributed import PartitionedTransport
    mesh, inputs3 = make_case(K)
    single = PartitionedTransport(mesh, inputs3, 0, 1)
    single_sweeps = [single.step(t, tol=1e-12, mass_flux=True, solver='jacobi').sweeps for t in range(3)]
    n = mesh['nreal'] + 1
    state = np.full((n, K), np.nan)
    for r in results:
        state[r[1]] = r[3]
    assert rel_err(state, single.gather_state()) <= 1e-10
    for r in results:
        got = [s for s, _ in r[6]]
        assert all(g <= s + 2 for g, s in zip(got[1:], single_sweeps[1:])), (got, single_sweeps)
        assert any(g % 2 == 1 for g in got), got          # N passes + ONE closing sweep (round 3: partitioned engines too)
    assert min(single_sweeps) >= 20                           # the case really iterates
    # cwr_step_info of a partitioned step: exchanges, how many of them overlapped, blocking checks.  A step of P J^2 passes at
    # halo depth d needs one exchange per d / 2 passes, one before the closing sweep (two with the even-passes shape), one in
    # front of the right-hand side (skipped when the previous step's tail delivered the rows) and the tail's own; in the steady state ONE check
    for r in results:
        for (sweeps, _), (exch, over, checks) in list(zip(r[6], r[13]))[1:]:
            passes = (sweeps - 1) // 2
            assert 0 < over <= exch <= -(-passes // max(1, depth // 2)) + 4, (sweeps, exch, over, depth)
            assert 1 <= checks <= 2, checks
        assert r[13] == results[0][13]                        # every rank made the same calls
    # the exchanges inside the pass loop ran beside the interior tiles (second stream + events), on every rank
    assert all(r[11] > 0 for r in results), [r[11] for r in results]


def _hip():
    """hipMemcpyAsync / hipStreamSynchronize of the HIP runtime the ENGINE is linked against (resolved through the engine
    library's own handle: a second copy of libamdhip64 in the process would not know the engine's stream)."""
    import ctypes
    import clearwater_riverine_amd as cw
    lib = cw.load_library()
    lib.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    lib.hipMemcpyAsync.restype = ctypes.c_int
    lib.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    lib.hipStreamSynchronize.restype = ctypes.c_int
    return lib


def _rank_hook(rank, world, K, route, uid_pipe, out_queue):
    """Three steps with the state rewritten between them -- through cwr_set_state, or through the device pointer of
    cwr_state_device_ptr fetched ONCE before the first step (what a device reaction kernel does)."""
    try:
        os.environ['CWR_RCCL_LIB'] = MOCK_LIB
        import clearwater_riverine_amd as cw
        from clearwater_riverine_amd.distributed import PartitionedTransport
        mesh, inputs3 = make_case(K)
        if rank == 0:
            uid = cw.TransportEngine.comm_unique_id()
            for _ in range(world - 1):
                uid_pipe.put(uid)
        else:
            uid = uid_pipe.get(timeout=120)
        pt = PartitionedTransport(mesh, inputs3, rank, world, device=0, unique_id=uid, halo_depth=6, renumber='hilbert')
        ptr, stream = pt.engine.state_device_ptr()
        hip = _hip()
        nc = pt.local.n_core
        for t in range(3):
            pt.step(t, tol=1e-12, mass_flux=True)            # (a mass-flux step leaves the halo rows "fresh")
            new = np.ascontiguousarray(pt.owned_state() * (1.0 + 0.1 * (t + 1)) + 0.25)
            if route == 'pointer':
                rc = hip.hipMemcpyAsync(ptr, new.ctypes.data, new.nbytes, 1, stream)
                rc2 = hip.hipStreamSynchronize(stream)
                if rc or rc2:
                    raise RuntimeError(f'hipMemcpyAsync -> {rc}, hipStreamSynchronize -> {rc2}')
            else:
                pt.engine.set_state(new)
        pt.step(3, tol=1e-12, mass_flux=True)
        out_queue.put((rank, pt.owned_reference_ids(), nc, pt.owned_state(), None, None, None, None))
        pt.engine.close()
    except Exception as exc:
        out_queue.put((rank, 0, 0, None, None, None, None, repr(exc)))


def test_state_rewritten_through_the_device_pointer_between_partitioned_steps(gpu_lib, monkeypatch):
    """the pointer of cwr_state_device_ptr never changes, so a caller fetches it once and writes through it
    between steps; after a mass-flux step the engine used to skip the next start-of-step halo exchange, and the replayed
    halo layers then built their right-hand sides from the pre-reaction state.  The pointer route must equal the
    cwr_set_state route bit for bit."""
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_TEST_BIG', '1')
    K = 4
    outs = {}
    for route in ('set_state', 'pointer'):
        results = run_ranks(2, _rank_hook, (K, route))
        mesh, _ = make_case(K)
        state = np.full((mesh['nreal'] + 1, K), np.nan)
        for r in results:
            state[r[1]] = r[3]
        assert not np.isnan(state).any()
        outs[route] = state
    assert np.array_equal(outs['set_state'], outs['pointer'])


@pytest.mark.parametrize('world,K,depth', [(1, 3, 1), (2, 3, 4), (3, 16, 8)])
def test_partitioned_real_cell_inputs_follow_the_reference(gpu_lib, world, K, depth, monkeypatch):
    """PartitionedTransport (the class bench.py and every multi-rank run go through) used to drop non-zero
    input_array entries on real cells at levels >= 1 (transport.py:258-264, linalg.py:199-200) without a word -- also with one
    rank.  Every rank now loads the entries of the cells it owns; state and fluxes must match the oracle."""
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_TEST_SOURCES', '1')
    mesh, inputs3 = make_case(K)
    n = mesh['nreal'] + 1
    assert np.count_nonzero(inputs3[1:, :n, :]) > 0
    state = np.full((n, K), np.nan)
    tot = np.full((len(mesh['edges_face1']), K), np.nan)
    if world == 1:
        from clearwater_riverine_amd.distributed import PartitionedTransport
        pt = PartitionedTransport(mesh, inputs3, 0, 1)
        for t in range(3):
            pt.step(t, tol=1e-12, mass_flux=True)
        state[:] = pt.gather_state()
        tot[:] = pt.engine.get_mass_flux()[2]
    else:
        for r in run_ranks(world, _rank_main, (K, 'auto', depth)):
            state[r[1]] = r[3]
            tot[r[4]] = r[5]
    oracle.derive_coefficients(mesh)
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for _ in range(3):
        ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[3, :n] for k in range(K)], axis=1)
    assert rel_err(state, want) <= 1e-9
    want_flux = np.stack([ref.constituent_dict[f'c{k}'].total_mass_flux[2] for k in range(K)], axis=1)
    assert flux_err(tot, want_flux) <= 1e-8


@pytest.mark.parametrize('world,K,depth', [(2, 16, 8), (4, 4, 6), (3, 1, 8)])
def test_overlapped_exchange_with_poisoned_halo_rows_under_the_asynchronous_stand_in(gpu_lib, world, K, depth, monkeypatch):
    """VERDICT r02 item 3.  Before every overlapped exchange the engine (test hook CWR_TEST_POISON_HALO) writes NaN into the
    halo rows of BOTH ping-pong vectors; the stand-in only enqueues on the engine's streams (CWR_MOCK_ASYNC=2: asynchronous or
    fail).  The result stays the bitwise single-rank result only if (i) the interior tiles, which run beside the exchange, read
    no halo row, (ii) the unpack on the communication stream waits for the pack (ev_packed), and (iii) the cut tiles on the
    engine's stream wait for the unpacked rows (ev_halo).  A missing wait shows up as NaN or as a stale row."""
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_TEST_BIG', '1')
    monkeypatch.setenv('CWR_TWO_CLOSING', '1')
    monkeypatch.setenv('CWR_LOCAL_REPS', '1')            # exact passes: bitwise independent of the partitioning
    monkeypatch.setenv('CWR_MOCK_ASYNC', '2')
    monkeypatch.setenv('CWR_TEST_POISON_HALO', '1')
    results = run_ranks(world, _rank_main, (K, 'jacobi', depth))
    assert all(r[12] == 1 for r in results), 'the stand-in fell back to its host-synchronous mode'
    assert all(r[11] > 0 for r in results), 'no exchange ran beside interior tiles'
    monkeypatch.delenv('CWR_TEST_POISON_HALO')
    from clearwater_riverine_amd.distributed import PartitionedTransport
    mesh, inputs3 = make_case(K)
    single = PartitionedTransport(mesh, inputs3, 0, 1)
    for t in range(3):
        single.step(t, tol=1e-12, mass_flux=True, solver='jacobi')
    n = mesh['nreal'] + 1
    state = np.full((n, K), np.nan)
    for r in results:
        state[r[1]] = r[3]
    assert not np.isnan(state).any()
    assert np.array_equal(state, single.gather_state())


@pytest.mark.parametrize('world,K,depth', [(2, 16, 8), (4, 4, 6)])
def test_closing_sweep_runs_its_core_tiles_beside_the_exchange_with_poisoned_halo_rows(gpu_lib, world, K, depth, monkeypatch):
    """Round 4: the exchange in front of the closing sweep of a partitioned step no longer runs alone on the
    engine's stream -- the sweep's core row tiles run beside it, its cut tiles and the replayed layers behind the unpack.  Default
    batch shape (one closing sweep), ping-pong passes, every halo row NaN before every overlapped exchange, asynchronous stand-in: the
    answer is the oracle's only if the core tiles read no halo row and the cut tiles wait for the unpacked values; the step counts
    one more overlapped exchange than with CWR_NO_CLOSING_OVERLAP=1, and the same exchanges and sweeps."""
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_TEST_BIG', '1')
    monkeypatch.setenv('CWR_MOCK_ASYNC', '2')
    monkeypatch.setenv('CWR_TEST_POISON_HALO', '1')
    split = run_ranks(world, _rank_main, (K, 'jacobi', depth))
    monkeypatch.setenv('CWR_NO_CLOSING_OVERLAP', '1')
    serial = run_ranks(world, _rank_main, (K, 'jacobi', depth))
    assert all(r[12] == 1 for r in split + serial), 'the stand-in fell back to its host-synchronous mode'
    mesh, inputs3 = make_case(K)
    n = mesh['nreal'] + 1
    oracle.derive_coefficients(mesh)
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for _ in range(3):
        ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[3, :n] for k in range(K)], axis=1)
    want_flux = np.stack([ref.constituent_dict[f'c{k}'].total_mass_flux[2] for k in range(K)], axis=1)
    states = []
    for results in (split, serial):
        state = np.full((n, K), np.nan)
        tot = np.full((len(mesh['edges_face1']), K), np.nan)
        for r in results:
            state[r[1]] = r[3]
            tot[r[4]] = r[5]
        assert not np.isnan(state).any()
        assert rel_err(state, want) <= 1e-9
        # the exchange that closes a step runs beside the fluxes of the faces between core cells (poisoned halo rows there too):
        # the fluxes of the cut faces must have waited for the unpacked rows
        assert flux_err(tot, want_flux) <= 1e-8
        states.append(state)
    assert np.array_equal(states[0], states[1])                  # the same arithmetic either way (ping-pong passes are deterministic)
    for a, b in zip(split, serial):
        assert a[6] == b[6]                                      # same sweeps
        for (ea, oa, ca), (eb, ob, cb) in zip(a[13], b[13]):
            # one closing sweep per batch (= per check): up to that many more overlapped exchanges -- a short follow-up batch (round 5:
            # eight sweeps behind a check that missed only the element-wise rule) may close without an exchange in front of its sweep
            assert ea == eb and ca == cb and 0 < oa - ob <= ca, (a[13], b[13])


def test_the_synchronous_mode_of_the_stand_in_still_works(gpu_lib, monkeypatch):
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_MOCK_ASYNC', '0')
    results = run_ranks(2, _rank_main, (3, 'auto', 4))
    assert all(r[12] == 0 for r in results)
    mesh, inputs3 = make_case(3)
    n = mesh['nreal'] + 1
    state = np.full((n, 3), np.nan)
    for r in results:
        state[r[1]] = r[3]
    oracle.derive_coefficients(mesh)
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(3)})
    for _ in range(3):
        ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[3, :n] for k in range(3)], axis=1)
    assert rel_err(state, want) <= 1e-9


@pytest.mark.parametrize('world,K,depth,grid', [(2, 16, 8, 32), (3, 16, 6, 16)])
def test_partitioned_engines_chain_their_tiles_too(gpu_lib, world, K, depth, grid, monkeypatch):
    """Round 3: a rank whose lists are long enough (here: the grid of the tiled pass capped at 32 / 16 blocks) links its tiles along
    the flow and relaxes in place between the halo exchanges, like a single engine.  Same answer as the oracle, fewer sweeps
    than the same partition with ping-pong passes, every rank with the same sweep counts."""
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_TEST_BIG', '1')
    monkeypatch.setenv('CWR_TCL_GRID', str(grid))
    monkeypatch.setenv('CWR_TILE_ORDER', 'lanes')
    chained = run_ranks(world, _rank_main, (K, 'jacobi', depth))
    assert all(r[14] for r in chained), 'no schedule was built on some rank'
    assert all(r[6] == chained[0][6] for r in chained)
    monkeypatch.setenv('CWR_NO_CHAINS', '1')
    plain = run_ranks(world, _rank_main, (K, 'jacobi', depth))
    assert not any(r[14] for r in plain)
    mesh, inputs3 = make_case(K)
    n = mesh['nreal'] + 1
    oracle.derive_coefficients(mesh)
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for _ in range(3):
        ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[3, :n] for k in range(K)], axis=1)
    for results in (chained, plain):
        state = np.full((n, K), np.nan)
        for r in results:
            state[r[1]] = r[3]
        assert rel_err(state, want) <= 1e-9
    sw_c = [s_ for s_, _ in chained[0][6]]
    sw_p = [s_ for s_, _ in plain[0][6]]
    assert sum(sw_c[1:]) < sum(sw_p[1:]), (sw_c, sw_p)
    # the in-loop exchanges of the chained ranks run beside their interior lists (separate schedules for interior and cut tiles;
    # the end ranks have interior tiles at this size, a middle rank of three may have none and then exchanges on its own stream)
    for r in (chained[0], chained[-1]):
        assert sum(o for _, o, _ in r[13]) > 0, r[13]


def test_ranks_that_cannot_all_chain_keep_the_ping_pong_passes_together(gpu_lib, monkeypatch):
    """Three ranks x 4 constituents (128-row tiles), grid capped at 16 blocks: the middle rank, with two halos, has 50 tiles and
    could chain (>= 3 per block: CWR_CHAIN_MIN_TILES=3 here), the end ranks with 45 cannot.  A chained rank always closes a batch with one plain sweep, a
    ping-pong rank chooses by the sweep count -- different exchanges per batch, i.e. a deadlock (seen with this very case).
    The ranks agree at their first solve: all chain or none."""
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_TEST_BIG', '1')
    monkeypatch.setenv('CWR_TCL_GRID', '16')
    monkeypatch.setenv('CWR_TILE_ORDER', 'lanes')
    monkeypatch.setenv('CWR_CHAIN_MIN_TILES', '3')            # (the threshold this case was built around; the default is 1.75 since round 4)
    K = 4
    results = run_ranks(3, _rank_main, (K, 'jacobi', 6))
    assert not any(r[14] for r in results), 'a rank chained although not every rank can'
    assert all(r[6] == results[0][6] for r in results)
    mesh, inputs3 = make_case(K)
    n = mesh['nreal'] + 1
    oracle.derive_coefficients(mesh)
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for _ in range(3):
        ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[3, :n] for k in range(K)], axis=1)
    state = np.full((n, K), np.nan)
    for r in results:
        state[r[1]] = r[3]
    assert rel_err(state, want) <= 1e-9


@pytest.mark.parametrize('world,K,depth,grid', [(2, 16, 8, 32), (3, 16, 6, 16)])
def test_chained_ranks_overlap_their_exchanges_and_survive_poisoned_halo_rows(gpu_lib, world, K, depth, grid, monkeypatch):
    """The poison test of the ping-pong passes for the chained ones: before every overlapped exchange all halo rows are NaN; the
    interior lists (in place, beside the exchange) must not read one, the lists of the cut tiles must wait for the unpacked rows.
    Chained passes are not bitwise reproducible, so the answer is checked against the oracle."""
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_TEST_BIG', '1')
    monkeypatch.setenv('CWR_TCL_GRID', str(grid))
    monkeypatch.setenv('CWR_TILE_ORDER', 'lanes')
    monkeypatch.setenv('CWR_MOCK_ASYNC', '2')
    monkeypatch.setenv('CWR_TEST_POISON_HALO', '1')
    results = run_ranks(world, _rank_main, (K, 'jacobi', depth))
    assert all(r[14] for r in results), 'no schedule was built on some rank'
    assert all(r[12] == 1 for r in results), 'the stand-in fell back to its host-synchronous mode'
    for r in (results[0], results[-1]):
        assert sum(o for _, o, _ in r[13]) > 0, r[13]
    mesh, inputs3 = make_case(K)
    n = mesh['nreal'] + 1
    oracle.derive_coefficients(mesh)
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for _ in range(3):
        ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[3, :n] for k in range(K)], axis=1)
    state = np.full((n, K), np.nan)
    for r in results:
        state[r[1]] = r[3]
    assert not np.isnan(state).any()
    assert rel_err(state, want) <= 1e-9


def _rank_ghost(rank, world, K, uid_pipe, out_queue):
    """A partitioned run whose flow field violates the reference's zero-coefficient precondition (linalg.py:349-351) at level 2,
    on faces that only ONE rank owns."""
    try:
        os.environ['CWR_RCCL_LIB'] = MOCK_LIB
        import clearwater_riverine_amd as cw
        from clearwater_riverine_amd.distributed import PartitionedTransport
        mesh, inputs3 = make_case(K)
        f2 = np.asarray(mesh['edges_face2'])
        inlet_faces = np.nonzero(np.isin(f2, mesh['inlet_ghost_cells']))[0]
        mesh['face_flow'] = mesh['face_flow'].copy()
        mesh['face_flow'][2, inlet_faces[:2]] = 0.0            # advection_coeff becomes 0 while the velocity stays < 0
        if rank == 0:
            uid = cw.TransportEngine.comm_unique_id()
            for _ in range(world - 1):
                uid_pipe.put(uid)
        else:
            uid = uid_pipe.get(timeout=120)
        pt = PartitionedTransport(mesh, inputs3, rank, world, device=0, unique_id=uid, halo_depth=4, renumber='hilbert')
        pt.step(0, tol=1e-12, mass_flux=True)
        before = pt.owned_state().copy()
        what = 'no error'
        try:
            pt.step(1, tol=1e-12, mass_flux=True)              # its right-hand side needs level 2
        except ValueError as exc:
            what = 'ValueError: ' + str(exc)
        except Exception as exc:                              # noqa: BLE001
            what = type(exc).__name__ + ': ' + str(exc)
        same = bool(np.array_equal(pt.owned_state(), before))
        owns = bool(np.isin(np.asarray(mesh['edges_face1'])[inlet_faces[:2]], pt.owned_reference_ids()).any())
        out_queue.put((rank, what, same, owns, None, None, None, None))
        pt.engine.close()
    except Exception as exc:
        out_queue.put((rank, None, None, None, None, None, None, repr(exc)))


def test_zero_coefficient_precondition_in_a_partitioned_run_is_raised_by_every_rank(gpu_lib, monkeypatch):
    """The violating rank NaN-poisons its right-hand side so that nobody is left in a collective; round 3: the flag travels with
    the all-reduced check, so EVERY rank raises the reference's ValueError (not just a NaN failure) and restores its state --
    without the second blocking download per step that used to fetch each rank's own counters."""
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    results = run_ranks(3, _rank_ghost, (2,))
    assert sum(r[3] for r in results) >= 1 and not all(r[3] for r in results)      # some ranks own the faces, some do not
    for r in results:
        assert r[1].startswith('ValueError') and 'ghost face' in r[1], r
        assert r[2], 'the failed step did not restore the state'


@pytest.mark.parametrize('seed', [int(v) for v in os.environ.get('CWR_SOAK_SEEDS', '').split(',') if v] or list(range(101, 113)))   # (CWR_SOAK_SEEDS=a,b,...: a longer hunt)
def test_partitioned_soak_random_meshes_worlds_and_depths(gpu_lib, seed, monkeypatch):
    """A seeded soak of the partitioned step through the stand-in (round 4; cf. tests/test_gpu_soak.py for one engine): random mesh
    (6- / 8-sided and dry cells, CFL 0.6 ... 25), 2-4 ranks, K, halo depth, numbering, grid caps that make the ranks chain or not, the
    one-stream or two-stream exchange -- three steps, every rank the same solver decisions, state against the oracle (max-norm 1e-9 and
    the element-wise bar), mass flux of the last step, the ledger against its own sum."""
    build_mock()
    rng = np.random.default_rng(seed)
    world = int(rng.choice([2, 3, 4]))
    K = int(rng.choice([1, 2, 4, 12, 16]))
    depth = int(rng.choice([1, 2, 4, 6, 8, 12]))
    monkeypatch.setenv('CWR_TEST_SOAK_SEED', str(seed))
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    if rng.random() < 0.7:
        monkeypatch.setenv('CWR_TCL_GRID', str(int(rng.choice([8, 16, 32]))))
    if rng.random() < 0.6:
        monkeypatch.setenv('CWR_TILE_ORDER', str(rng.choice(['lanes', 'hilbert'])))
    if rng.random() < 0.3:
        monkeypatch.setenv('CWR_CHAIN_MIN_TILES', '1')
    if rng.random() < 0.25:
        monkeypatch.setenv('CWR_COMM_TWO_STREAMS', '1')
    if rng.random() < 0.25:
        monkeypatch.setenv('CWR_NO_CLOSING_OVERLAP', '1')
    results = run_ranks(world, _rank_main, (K, 'auto', depth))
    mesh, inputs3 = make_case(K)
    n = mesh['nreal'] + 1
    state = np.full((n, K), np.nan)
    for r in results:
        state[r[1]] = r[3]
    assert not np.isnan(state).any() and sum(len(r[1]) for r in results) == n
    assert all(r[6] == results[0][6] for r in results), [r[6] for r in results]          # the same sweeps / iterations on every rank
    oracle.derive_coefficients(mesh)
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for _ in range(3):
            ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[3, :n] for k in range(K)], axis=1)
    assert rel_err(state, want) <= 1e-9
    tot = np.full((len(mesh['edges_face1']), K), np.nan)
    for r in results:
        tot[r[4]] = r[5]
    want_flux = np.stack([ref.constituent_dict[f'c{k}'].total_mass_flux[2] for k in range(K)], axis=1)
    assert flux_err(tot, want_flux) <= 1e-8


@pytest.mark.parametrize('world,K,depth,grid', [(2, 16, 8, 32), (3, 16, 6, 16), (3, 4, 8, 16)])
def test_deterministic_steps_of_chained_ranks_walk_their_lists_and_repeat_bit_for_bit(gpu_lib, world, K, depth, grid, monkeypatch):
    """Round 4: CWR_STEP_DETERMINISTIC on ranks that chain.  The passes go from one vector into the other -- along the same lists, cut
    into interior and cut tiles for the passes with an exchange, a tile taking its predecessor's rows from LDS -- so nothing depends on
    timing: two runs agree bit for bit (cwr_step_info.chained = 2 on every rank), the exchanges still run beside the interior lists
    (poisoned halo rows), and the answer is the oracle's."""
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_TEST_BIG', '1')
    monkeypatch.setenv('CWR_TCL_GRID', str(grid))
    monkeypatch.setenv('CWR_TILE_ORDER', 'lanes')
    monkeypatch.setenv('CWR_TEST_POISON_HALO', '1')
    monkeypatch.setenv('CWR_TEST_DETERMINISTIC', '1')
    runs = [run_ranks(world, _rank_main, (K, 'jacobi', depth)) for _ in range(2)]
    for results in runs:
        assert all(r[14] for r in results), 'no schedule was built on some rank'
        assert all(k == 2 for r in results for k in r[15]), [r[15] for r in results]
        assert all(r[6] == results[0][6] for r in results)
        for r in (results[0], results[-1]):
            assert sum(o for _, o, _ in r[13]) > 0, r[13]            # overlapped exchanges
    for a, b in zip(*runs):
        assert np.array_equal(a[3], b[3]) and a[6] == b[6]              # bit for bit, the same sweeps
    mesh, inputs3 = make_case(K)
    n = mesh['nreal'] + 1
    oracle.derive_coefficients(mesh)
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for _ in range(3):
        ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[3, :n] for k in range(K)], axis=1)
    state = np.full((n, K), np.nan)
    for r in runs[0]:
        state[r[1]] = r[3]
    assert rel_err(state, want) <= 1e-9
    # the same ranks without the flag: in place at more than 8 constituents (chained = 1); up to 8 the deterministic passes are the default
    monkeypatch.delenv('CWR_TEST_DETERMINISTIC')
    plain = run_ranks(world, _rank_main, (K, 'jacobi', depth))
    assert all(k == (1 if K > 8 else 2) for r in plain for k in r[15]), [r[15] for r in plain]


# ---------------------------------------------------------------------------------------------------------------------------------
# round 5: the row-wise error factor of partitioned engines (VERDICT r04 task 3)
def _dry_case(K):
    """The 30 %-dry-cell mesh of test_gpu_robustness.test_a_third_of_the_cells_dry_... (linalg.py:66,76-81: dry cells get a dummy
    diagonal; their wet neighbours have row sums above 1, so ||J||_inf gives no bound at all)."""
    import clearwater_riverine_amd as cw
    nx, ny, steps = 90, 40, 4
    mesh = cw.synthetic.make_mesh(nx=nx, ny=ny, n_steps=steps, seed=12, n_merge=nx * ny // 25, n_dry=int(0.3 * nx * ny), dt=30.0,
                                  diffusion_coefficient=0.5)
    return mesh, cw.synthetic.distinct_input_array(mesh, K, seed=12), steps


def _rank_dry(rank, world, K, depth, uid_pipe, out_queue):
    try:
        os.environ['CWR_RCCL_LIB'] = MOCK_LIB
        import warnings
        import clearwater_riverine_amd as cw
        from clearwater_riverine_amd.distributed import PartitionedTransport
        if rank == 0:
            uid = cw.TransportEngine.comm_unique_id()
            for _ in range(world - 1):
                uid_pipe.put(uid)
        else:
            uid = uid_pipe.get(timeout=120)
        mesh, inputs3, steps = _dry_case(K)
        pt = PartitionedTransport(mesh, inputs3, rank, world, device=0, unique_id=uid, halo_depth=depth, renumber='hilbert')
        F, jn = pt.engine.error_factors(), pt.engine.jacobi_norms()
        flags = []
        with warnings.catch_warnings():
            warnings.simplefilter('error')                     # (a clamp would warn)
            for t in range(steps):
                flags.append(pt.step(t, tol=1e-12, max_iter=20000).flags)
        out_queue.put((rank, pt.owned_reference_ids(), pt.owned_state(), F, jn, flags, None, None))
        pt.engine.close()
    except Exception as exc:
        out_queue.put((rank, None, None, None, None, None, None, repr(exc)))


@pytest.mark.parametrize('world,depth', [(2, 4), (4, 6)])
def test_partitioned_engines_hold_the_row_wise_error_factor_of_the_global_matrix(gpu_lib, world, depth, monkeypatch):
    """spsolve (transport.py:249) is exact on any partition; until round 5 a partitioned engine kept the NORM form of the error factor
    (||J||_inf / (1 - ||J||_inf): infinite beside a dry cell), so the same mesh ran without flags on one GPU and clamped + warned on N.
    Now the Neumann sweeps of refine_error_factors run over the ranks with one halo exchange per `depth` sweeps and one all-reduce per
    check: every rank holds the single engine's factors, no rank sets a flag, and the result is the oracle's element by element."""
    from test_gpu_parity import make_engine
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    K = 3
    results = run_ranks(world, _rank_dry, (K, depth))
    mesh, inputs3, steps = _dry_case(K)
    oracle.derive_coefficients(mesh)
    n = mesh['nreal'] + 1
    single = make_engine(mesh, inputs3)
    F1, jn1 = single.error_factors(), single.jacobi_norms()
    single.close()
    assert jn1[:steps].max() > 1.0 and F1[:steps].max() < 300.0
    state = np.full((n, K), np.nan)
    for r in results:
        rank, ids, st, F, jn, flags = r[:6]
        assert flags == [0] * steps, (rank, flags)
        assert np.allclose(jn[:steps], jn1[:steps], rtol=1e-12)                    # the maximum over the ranks = the single engine's
        # the same sweeps of the same rows with the same stop decisions: the single engine's factors (rounding of the row sums aside)
        assert np.allclose(F[:steps], F1[:steps], rtol=1e-9), (rank, F[:steps], F1[:steps])
        state[ids] = st
    assert not np.isnan(state).any()
    from util import oracle_run
    ref = oracle_run(mesh, inputs3, steps)
    for k in range(K):
        assert rel_err(state[:, k], ref.constituent_dict[f'c{k}'].state[steps, :n]) <= 1e-9


def _stiff_band(K):
    import clearwater_riverine_amd as cw
    T = 34
    mesh = cw.synthetic.make_mesh(300, 60, T, seed=20100529, n_merge=0, dx=75.0, dy=75.0, depth=3.0, dt=3600.0, velocity=0.3, breathing=0.0,
                                  diffusion_coefficient=0.1, period_steps=24)
    return mesh, cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=86400.0)


def _rank_stiff(rank, world, K, steps, uid_pipe, out_queue):
    try:
        os.environ['CWR_RCCL_LIB'] = MOCK_LIB
        import clearwater_riverine_amd as cw
        from clearwater_riverine_amd.distributed import PartitionedTransport
        if rank == 0:
            uid = cw.TransportEngine.comm_unique_id()
            for _ in range(world - 1):
                uid_pipe.put(uid)
        else:
            uid = uid_pipe.get(timeout=120)
        mesh, inputs3 = _stiff_band(K)
        pt = PartitionedTransport(mesh, inputs3, rank, world, device=0, unique_id=uid)
        info = []
        for t in range(steps):
            r = pt.step(t, tol=1e-12, mass_flux=True)
            info.append((r.sweeps, r.iterations, r.flags, r.checks))
        out_queue.put((rank, pt.owned_reference_ids(), 0, pt.owned_state(), None, None, info, None))
        pt.engine.close()
    except Exception as exc:
        out_queue.put((rank, 0, 0, None, None, None, None, repr(exc)))


def test_partitioned_ranks_at_cfl_18_keep_their_batches_in_hand(gpu_lib, monkeypatch):
    """The run-away of the passes' batch-size prediction (round 5; test_gpu_behaviour.py) on a partition: the 18 k-cell band at the
    reference's own time step, K = 4, 2 ranks through the stand-in, 30 steps -- every rank the same sweeps and checks (the check is
    all-reduced), no BiCGSTAB, no step beyond 1.6 x the median, and the single engine's answer."""
    build_mock()
    from clearwater_riverine_amd.distributed import PartitionedTransport
    K, steps, world = 4, 30, 2
    results = run_ranks(world, _rank_stiff, (K, steps))
    assert all(r[6] == results[0][6] for r in results)
    sweeps = [i[0] for i in results[0][6]]
    assert all(i[1] == 0 and i[2] == 0 for i in results[0][6]), results[0][6]
    assert max(sweeps[1:]) <= 1.6 * float(np.median(sweeps)), sweeps
    mesh, inputs3 = _stiff_band(K)
    pt = PartitionedTransport(mesh, inputs3, 0, 1)
    for t in range(steps):
        pt.step(t, tol=1e-12, mass_flux=True)
    want = pt.gather_state()
    n = mesh['nreal'] + 1
    got = np.full((n, K), np.nan)
    for r in results:
        got[r[1]] = r[3]
    pt.engine.close()
    assert not np.isnan(got).any()
    assert np.max(np.abs(got - want[:n])) <= 1e-9 * np.max(np.abs(want[:n]))


@pytest.mark.parametrize('world,K,depth,W,big,lazy', [(2, 3, 4, 2, False, False), (4, 16, 8, 3, False, False), (2, 4, 8, 2, True, False), (4, 1, 6, 4, True, False),
                                                       (2, 3, 4, 4, False, True), (4, 4, 8, 2, True, True)])
def test_windowed_ranks_equal_resident_ranks_bit_for_bit(gpu_lib, world, K, depth, W, big, lazy, monkeypatch):
    """cwr_flow_window_open / _load on PARTITIONED engines.  Every rank keeps a ring of W levels of its slices; the
    zero-coefficient flag and ||J||_inf of an arriving level are all-reduced on the communication stream where the level is loaded, the
    row-wise error factor is taken -- collectively -- where its step runs (the small mesh has dry cells: no norm bound).  Through the
    stream-asynchronous stand-in, deterministic passes: states, fluxes, sweep counts and the ledger of every rank equal to the same ranks
    with every level resident, BIT FOR BIT; the exchanges still run beside compute."""
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    monkeypatch.setenv('CWR_TEST_DETERMINISTIC', '1')
    monkeypatch.setenv('CWR_MOCK_ASYNC', '2')
    if big:
        monkeypatch.setenv('CWR_TEST_BIG', '1')
    resident = run_ranks(world, _rank_main, (K, 'jacobi', depth))
    monkeypatch.setenv('CWR_TEST_FLOW_WINDOW', str(W))
    if lazy:                                                    # (the ranks cut their slices out of a level source over the whole mesh: levels.FlowWindowFeeder)
        monkeypatch.setenv('CWR_TEST_LAZY', '1')
    windowed = run_ranks(world, _rank_main, (K, 'jacobi', depth))
    for a, b in zip(resident, windowed):
        assert a[0] == b[0] and np.array_equal(a[1], b[1])
        assert a[6] == b[6], (a[0], a[6], b[6])                                  # sweeps / iterations of every step
        assert np.array_equal(a[3], b[3], equal_nan=True), f'rank {a[0]}: windowed state differs from the resident one'
        assert np.array_equal(a[5], b[5], equal_nan=True)                        # total mass flux of the faces the rank reports
        assert np.array_equal(a[8], b[8], equal_nan=True)                        # boundary-line ledger
        assert a[13] == b[13] and b[12] == 1                                     # the same exchanges / overlaps / checks; asynchronous stand-in
    # ... and the answer is the oracle's
    mesh, inputs3 = make_case(K)
    n = mesh['nreal'] + 1
    state = np.full((n, K), np.nan)
    for r in windowed:
        state[r[1]] = r[3]
    oracle.derive_coefficients(mesh)
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for _ in range(3):
        ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[3, :n] for k in range(K)], axis=1)
    assert rel_err(state, want) <= 1e-9


def _rank_groups(rank, world, K, G, depth, uid_pipe, out_queue):
    """One rank of R cell ranges x G constituent groups (distributed.GroupedTransport): three steps, then its rows of its columns."""
    try:
        os.environ['CWR_RCCL_LIB'] = MOCK_LIB
        import clearwater_riverine_amd as cw
        from clearwater_riverine_amd.distributed import GroupedTransport, group_layout
        if rank == 0:
            ids = [cw.TransportEngine.comm_unique_id() for _ in range(G)]      # one communicator per group
            for _ in range(world - 1):
                uid_pipe.put(ids)
        else:
            ids = uid_pipe.get(timeout=120)
        g, r, R, k0, k1 = group_layout(rank, world, G, K)
        mesh, inputs3 = make_case(K)
        gt = GroupedTransport(mesh, inputs3, rank, world, G, device=0, unique_id=ids[g] if R > 1 else None, halo_depth=depth,
                              renumber='hilbert' if depth >= 4 else None)
        infos = []
        for t in range(3):
            res = gt.step(t, tol=1e-12, mass_flux=True)
            infos.append((res.sweeps, res.iterations, res.exchanges))
        out_queue.put((rank, gt.owned_reference_ids(), (g, r, k0, k1), gt.owned_state(), None, None, infos, None, len(gt.local.peers)))
        gt.engine.close()
    except Exception as exc:
        out_queue.put((rank, None, None, None, None, None, None, repr(exc), 0))


@pytest.mark.parametrize('world,K,G,depth', [(4, 4, 2, 4), (4, 6, 2, 2), (4, 16, 4, 8), (3, 3, 1, 4)])    # (<= 5 rank processes: the runner holds the GPU too, 6 per card)
def test_constituent_groups_times_cell_ranges_match_the_oracle(gpu_lib, world, K, G, depth, monkeypatch):
    """Round 6: N ranks as N / G contiguous cell ranges x G groups of constituents.  A group is a complete partitioned
    run of its constituents with a communicator of its own (the K systems share A and never talk to each other, transport.py:231-249):
    the ranks of a group exchange halos among themselves and with nobody else.  Assembled from all ranks -- rows by range, columns by group --
    the state is the oracle's; every rank of a group takes the same solver decisions; a rank's peers are ranks of its own group only
    (G = 4 of 4 ranks: no communicator at all; G = 1: plain PartitionedTransport)."""
    build_mock()
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    results = run_ranks(world, _rank_groups, (K, G, depth))
    mesh, inputs3 = make_case(K)
    n = mesh['nreal'] + 1
    R = world // G
    state = np.full((n, K), np.nan)
    for r in results:
        g, rr, k0, k1 = r[2]
        assert (g, rr) == divmod(r[0], R) and r[3].shape[1] == k1 - k0
        state[np.asarray(r[1])[:, None], np.arange(k0, k1)[None, :]] = r[3]
        assert r[8] <= R - 1                                         # peers within the group's ranges only
        if R > 1:
            assert all(i[2] > 0 for i in r[6])                       # ... and the group's ranks did exchange
    assert not np.isnan(state).any()
    for g in range(G):                                               # the ranks of one group decide together
        grp = [r for r in results if r[2][0] == g]
        assert all(r[6] == grp[0][6] for r in grp)
    oracle.derive_coefficients(mesh)
    ref = oracle.OracleModel(mesh, {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for _ in range(3):
        ref.update()
    want = np.stack([ref.constituent_dict[f'c{k}'].state[3, :n] for k in range(K)], axis=1)
    assert rel_err(state, want) <= 1e-9
