"""The engine's host-side index builders (csrc/cwr_host_builders.hpp) on the CPU, under AddressSanitizer + UBSan with
bounds-checked std::vector access.

A wrong index here is an out-of-range LDS / global access in k_sq_tiled or k_sq_numeric on the GPU; the one process fault of
round 3 was an out-of-bounds read of an (empty) host vector in exactly this code (profiles/r04_b_exit_fault_forensics.txt).  The
driver tests/host_builders/builders_main.cpp is compiled here with g++ -fsanitize=address,undefined -D_GLIBCXX_ASSERTIONS, fed
random meshes (merged 5-8-sided cells, dry cells, shuffled numberings, single engines and ranks of a partition with deep halos)
and its results are compared with numpy statements of the same constructions: the two-hop pattern in discovery order, the
per-tile column lists and 16-bit positions, the tile links, the chains and per-block lists (schedule.py) and the carry-over
codes of consecutive tiles."""
import os
import struct
import subprocess

import numpy as np
import pytest

import clearwater_riverine_amd as cw
from clearwater_riverine_amd import schedule as sch
from clearwater_riverine_amd.ordering import hilbert_order, lane_order, renumber_mesh
from clearwater_riverine_amd.partition import partition_mesh

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'host_builders', 'builders_main.cpp')
HDR = os.path.join(os.path.dirname(HERE), 'clearwater-riverine_amd', 'csrc', 'cwr_host_builders.hpp')


@pytest.fixture(scope='module')
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('hb') / 'builders_main')
    subprocess.run(['g++', '-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                    '-D_GLIBCXX_ASSERTIONS', '-Wall', '-Werror', SRC, '-o', exe], check=True)
    return exe


def write_bag(path, bag):
    with open(path, 'wb') as fh:
        fh.write(struct.pack('<i', len(bag)))
        for name, arr in bag.items():
            a = np.ascontiguousarray(arr)
            if a.dtype == np.float32:
                a = a.view(np.int32)
            a = a.astype('<i4')
            fh.write(struct.pack('<i', len(name))); fh.write(name.encode()); fh.write(struct.pack('<i', a.size)); fh.write(a.tobytes())


def read_bag(path):
    raw = open(path, 'rb').read()
    off, bag = 4, {}
    for _ in range(struct.unpack_from('<i', raw, 0)[0]):
        ln = struct.unpack_from('<i', raw, off)[0]; off += 4
        name = raw[off:off + ln].decode(); off += ln
        cnt = struct.unpack_from('<i', raw, off)[0]; off += 4
        bag[name] = np.frombuffer(raw, dtype='<i4', count=cnt, offset=off).copy(); off += 4 * cnt
    return bag


def adjacency(f1, f2, n_owned, n_real):
    """CSR adjacency as cwr_create builds it: per computed row its (cell, face) entries in ascending face id; neighbour id, or
    -1 - ghost; edge code = face << 1 | side (identity face order here)."""
    f1 = np.asarray(f1, dtype=np.int64); f2 = np.asarray(f2, dtype=np.int64)
    e = np.arange(len(f1), dtype=np.int64)
    cell = np.concatenate([f1, f2])
    code = np.concatenate([e << 1, (e << 1) | 1])
    nb = np.concatenate([np.where(f2 < n_real, f2, -1 - (f2 - n_real)), f1])
    keep = cell < n_owned
    cell, code, nb = cell[keep], code[keep], nb[keep]
    o = np.lexsort((code, cell))
    ptr = np.zeros(n_owned + 1, dtype=np.int64)
    np.add.at(ptr, cell + 1, 1)
    return np.cumsum(ptr).astype(np.int32), nb[o].astype(np.int32), code[o].astype(np.int32)


def run(driver, tmp_path, ptr, nb, edge, adv, n_owned, n_core, n_real, K, tr, grid, seg=1 << 20, nvmax=0, limits=()):
    fin, fout = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
    write_bag(fin, {'params': np.array([n_owned, n_core, n_real, K, tr, grid, seg, nvmax, *limits]), 'ptr': ptr, 'nb': nb, 'edge': edge,
                    'adv': np.asarray(adv, dtype=np.float32)})
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=1', UBSAN_OPTIONS='print_stacktrace=1')
    res = subprocess.run([driver, fin, fout], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0 and res.stderr == '', (res.returncode, res.stderr[-3000:])
    return read_bag(fout)


def check_pattern(out, ptr, nb, n_owned, n_core):
    n = int(out['n_sq'][0])
    # rows with a J^2 row: the longest prefix whose real neighbours all have rows of their own
    want_n = n_owned
    for c in range(n_owned):
        if (nb[ptr[c]:ptr[c + 1]] >= n_owned).any():
            want_n = c
            break
    assert n == want_n >= n_core
    ptr2, col2, pair_ptr, slots = out['ptr2'], out['col2'], out['pair_ptr'], out['slots']
    longest = 0
    for c in range(n):
        cols, sl = [], []
        for j in range(ptr[c], ptr[c + 1]):
            m = nb[j]
            if m < 0:
                continue
            for i in range(ptr[m], ptr[m + 1]):
                k = nb[i]
                if k < 0:
                    continue
                if k not in cols:
                    cols.append(k)
                sl.append(cols.index(k))
        assert list(col2[ptr2[c]:ptr2[c + 1]]) == cols, c
        assert list(slots[pair_ptr[c]:pair_ptr[c + 1]]) == sl, c
        longest = max(longest, len(cols))
    assert int(out['n_sq'][1]) == longest and len(col2) == ptr2[n] and len(slots) == pair_ptr[n]
    # the branch-free path of k_sq_numeric reads neighbour rows as if their r-th entry were their r-th product
    deg = np.diff(ptr)
    DEG = 4 if deg.max() <= 4 else (6 if deg.max() <= 6 else 8)
    ghosty = np.array([(nb[ptr[c]:ptr[c + 1]] < 0).any() for c in range(n_owned)])
    for c in np.nonzero(out['fast'])[0]:
        ms = [m for m in nb[ptr[c]:ptr[c + 1]] if m >= 0]
        assert 0 < deg[c] <= DEG and ms and all(deg[m] <= DEG and not ghosty[m] for m in ms)
    return n


def check_tiling(out, n, K, tr, n_real):
    ptr2, col2 = out['ptr2'], out['col2']
    trow, tptr, tcols, loc2 = out['trow'], out['tptr'], out['tcols'], out['loc2']
    nt = len(trow) - 1
    assert list(trow) == list(range(0, n, tr)) + [n]                      # fixed-size tiles
    max_cols = cap = 0
    for t in range(nt):
        c0, c1 = trow[t], trow[t + 1]
        ent = col2[ptr2[c0]:ptr2[c1]]
        others = np.setdiff1d(np.unique(ent), np.arange(c0, c1))
        want = np.concatenate([np.arange(c0, c1), others])
        got = tcols[tptr[t]:tptr[t + 1]]
        assert np.array_equal(got, want), t
        assert got.min() >= 0 and got.max() < n_real
        pos = {int(g): i for i, g in enumerate(want)}
        assert np.array_equal(loc2[ptr2[c0]:ptr2[c1]], [pos[int(k)] * K for k in ent]), t
        max_cols, cap = max(max_cols, len(want)), max(cap, len(ent))
    assert list(out['tile_dims']) == [max_cols, cap + (cap & 1), nt] and max_cols * K <= 65535
    assert len(loc2) == len(col2)
    # meta: per tile its rows' ptr2 entries (fixed tiles have no virtual items)
    assert np.array_equal(out['meta'], ptr2[:n]) and len(out['vtab']) == 0
    return nt


def check_chains(out, f1, f2, adv, n, tr, nt, grid, n_real, n_core):
    # links: every ordered pair of distinct tiles that share a face between rows with a J^2 row, sorted by (src, dst)
    f1 = np.asarray(f1, dtype=np.int64); f2 = np.asarray(f2, dtype=np.int64)
    ok = (f1 < n) & (f2 < n) & (f1 // tr != f2 // tr)
    a, b = f1[ok] // tr, f2[ok] // tr
    keys = np.unique(np.concatenate([a * nt + b, b * nt + a]))
    assert np.array_equal(out['link_src'].astype(np.int64) * nt + out['link_dst'], keys)
    flux = out['link_flux'].view(np.float32).astype(np.float64)
    us, ud, w = sch.tile_links(f1, f2, adv, n, tr, nt)                     # the numpy statement (links with flow only)
    live = flux > 0
    assert np.array_equal(out['link_src'][live].astype(np.int64) * nt + out['link_dst'][live], us * nt + ud)
    assert np.allclose(flux[live], w, rtol=1e-6)
    # chains from the driver's own float sums (so that ties fall the same way), then lists: schedule.py is the specification
    chains = sch.chains(out['link_src'][live].astype(np.int64), out['link_dst'][live].astype(np.int64), flux[live], nt)
    nxt = np.full(nt, -1, dtype=np.int64)
    for ch in chains:
        nxt[ch[:-1]] = ch[1:]
    assert np.array_equal(out['nxt'], nxt)
    for spb in (1, 2):
        depth = int(out[f'depth{spb}'][0])
        got = out[f'sched{spb}'].reshape(depth, grid)
        assert np.array_equal(got, sch.schedule(chains, nt, grid, streams_per_block=spb))
    # carry-over codes along every list: -2 - (position in the previous tile's column list) exactly for the shared columns
    tptr, tcols = out['tptr'], out['tcols']
    sched = out['sched1'].reshape(int(out['depth1'][0]), grid)
    check_codes(out['scols'], [sched], tptr, tcols)
    if 'sched_in' in out:
        gi, d_in, go, d_out = (int(v) for v in out['sub_dims'])
        s_in, s_out = out['sched_in'].reshape(d_in, gi), out['sched_out'].reshape(d_out, go)
        inner, outer = set(out['inner'].tolist()), set(out['outer'].tolist())
        assert inner | outer == set(range(nt)) and not (inner & outer)
        assert sorted(s_in[s_in >= 0].tolist()) == sorted(inner) and sorted(s_out[s_out >= 0].tolist()) == sorted(outer)
        for t in inner:                                                     # an interior tile touches core rows only
            assert out['trow'][t + 1] <= n_core and (tcols[tptr[t]:tptr[t + 1]] < n_core).all()
        for s, members in ((s_in, inner), (s_out, outer)):                  # a chain link inside one of the two sets stays a list neighbour
            follows = {int(p): int(q) for bcol in s.T for p, q in zip(bcol[bcol >= 0][:-1], bcol[bcol >= 0][1:])}
            firsts = {int(bcol[0]) for bcol in s.T if bcol[0] >= 0}
            for p in members:
                if nxt[p] >= 0 and int(nxt[p]) in members:
                    assert follows.get(int(p)) == int(nxt[p]) or int(nxt[p]) in firsts, (p, nxt[p])   # (or the list was cut there)
        check_codes(out['scols_io'], [s_in, s_out], tptr, tcols)


def check_codes(scols, scheds, tptr, tcols):
    want = tcols.copy()
    for sched in scheds:
        for bcol in sched.T:
            lst = bcol[bcol >= 0]
            assert (bcol[:len(lst)] >= 0).all()                             # dense from the top
            for prev, cur in zip(lst[:-1], lst[1:]):
                pcols = tcols[tptr[prev]:tptr[prev + 1]]
                where = {int(g): i for i, g in enumerate(pcols)}
                for q in range(tptr[cur], tptr[cur + 1]):
                    if int(tcols[q]) in where:
                        want[q] = -2 - where[int(tcols[q])]
    assert np.array_equal(scols, want)
    assert (scols <= -2).any() or all(len(bcol[bcol >= 0]) < 2 for sched in scheds for bcol in sched.T)      # (the case really carries columns over)
    # what the kernel relies on: a coded position fits 16 bits below the 0xFFFF marker
    coded = scols[scols <= -2]
    assert len(coded) == 0 or (-2 - coded).max() < 0xFFFF


CASES = [  # nx, ny, seed, merges, dry, K (-> tile rows), grid, order
    (40, 24, 1, 40, 0, 16, 8, 'lanes'),
    (57, 31, 2, 120, 3, 1, 8, 'hilbert'),
    (64, 20, 3, 0, 0, 12, 16, 'lanes'),
    (33, 33, 4, 150, 5, 4, 8, 'shuffled'),
    (90, 12, 5, 60, 1, 8, 24, 'lanes'),
]


@pytest.mark.parametrize('nx,ny,seed,n_merge,n_dry,K,grid,order', CASES)
def test_single_engine_builders_against_numpy_statements(driver, tmp_path, nx, ny, seed, n_merge, n_dry, K, grid, order):
    mesh = cw.synthetic.make_mesh(nx, ny, 3, seed=seed, n_merge=n_merge, n_dry=n_dry, shuffle_window=16 if order == 'shuffled' else 0,
                                  dt=40.0, diffusion_coefficient=0.5)
    n = mesh['nreal'] + 1
    tr = {16: 64, 12: 85, 8: 128, 4: 128, 1: 256}[K]
    if order == 'lanes':
        mesh = renumber_mesh(mesh, lane_order(mesh, n, tile_rows=tr))
    elif order == 'hilbert':
        mesh = renumber_mesh(mesh, hilbert_order(mesh['face_x'], mesh['face_y'], n))
    f1, f2 = np.asarray(mesh['edges_face1']), np.asarray(mesh['edges_face2'])
    adv = np.asarray(mesh['face_flow'][1], dtype=np.float32)
    ptr, nb, edge = adjacency(f1, f2, n, n)
    out = run(driver, tmp_path, ptr, nb, edge, adv, n, n, n, K, tr, grid)
    assert out['sq_ok'][0] == 1 and out['tiled'][0] == 1
    assert check_pattern(out, ptr, nb, n, n) == n
    nt = check_tiling(out, n, K, tr, n)
    check_chains(out, f1, f2, adv, n, tr, nt, grid, n, n)


@pytest.mark.parametrize('world,rank,depth,K', [(3, 1, 6, 16), (8, 3, 8, 16), (8, 0, 4, 1), (2, 1, 1, 4)])
def test_builders_on_a_rank_of_a_partition(driver, tmp_path, world, rank, depth, K):
    """Halo rows: the J^2 rows stop at the first row with a neighbour outside the computed rows (depth 1: no J^2 at all), column
    lists reach into the halo, and the interior / cut tiles get schedules of their own over one shared copy of the codes."""
    mesh = cw.synthetic.make_mesh(120, 64, 3, seed=7, n_merge=300, dt=40.0, diffusion_coefficient=0.5)
    n = mesh['nreal'] + 1
    tr = {16: 64, 4: 128, 1: 256}[K]
    mesh = renumber_mesh(mesh, lane_order(mesh, n, tile_rows=tr))
    lm = partition_mesh(mesh['edges_face1'], mesh['edges_face2'], n, world, rank, depth=depth, align=tr)
    n_owned, n_real = lm.n_rows, lm.n_rows + lm.n_halo
    ptr, nb, edge = adjacency(lm.face1, lm.face2, n_owned, n_real)
    adv = np.asarray(mesh['face_flow'][1], dtype=np.float32)[lm.edge_global]
    out = run(driver, tmp_path, ptr, nb, edge, adv, n_owned, lm.n_core, n_real, K, tr, 8)
    if depth < 2:
        assert out['sq_ok'][0] == 0                                         # halo too shallow for two sweeps per launch
        return
    assert out['sq_ok'][0] == 1 and out['tiled'][0] == 1
    n_sq = check_pattern(out, ptr, nb, n_owned, lm.n_core)
    assert lm.n_core <= n_sq <= n_owned
    nt = check_tiling(out, n_sq, K, tr, n_real)
    check_chains(out, lm.face1, lm.face2, adv, n_sq, tr, nt, 8, n_real, lm.n_core)


def test_work_item_tiles_cover_every_row_once(driver, tmp_path):
    """seg = 12 (the work-item build): long rows occupy one lane-group slot per chunk, tiles hold a variable number of rows."""
    mesh = cw.synthetic.make_mesh(48, 40, 3, seed=11, n_merge=400)
    n = mesh['nreal'] + 1
    f1, f2 = np.asarray(mesh['edges_face1']), np.asarray(mesh['edges_face2'])
    ptr, nb, edge = adjacency(f1, f2, n, n)
    out = run(driver, tmp_path, ptr, nb, edge, np.zeros(len(f1), np.float32), n, n, n, 1, 256, 8, seg=12, nvmax=48)
    assert out['tiled'][0] == 1
    trow, vptr, vtab, ptr2 = out['trow'], out['vptr'], out['vtab'], out['ptr2']
    assert trow[0] == 0 and trow[-1] == n and (np.diff(trow) > 0).all()
    for t in range(len(trow) - 1):
        rows = trow[t + 1] - trow[t]
        lens = np.diff(ptr2[trow[t]:trow[t + 1] + 1])
        extra = np.maximum(0, (lens - 1) // 12)
        assert rows + extra.sum() <= 256 and extra.sum() <= 48 and vptr[t + 1] - vptr[t] == extra.sum()
        codes = vtab[vptr[t]:vptr[t + 1]]
        want = [r | (ch << 8) for r in range(rows) for ch in range(1, extra[r] + 1)]
        assert list(codes) == want


def test_heavy_windows_are_cut_into_smaller_tiles_and_the_rest_keep_their_windows(driver, tmp_path):
    """Round 5: a middle rank of a partition has windows of 64 rows in which the replayed strips of two neighbours meet -- two clusters,
    two neighbourhoods, more distinct x rows than the cheapest kernel configuration holds.  With limits build_tiling cuts such a window
    into halves (recursively); every other tile is its whole window, every piece respects the limits, the links follow the row ranges
    (ASan / UBSan build) and the schedules over the cut tiling are valid (checked by the driver itself)."""
    mesh = cw.synthetic.make_mesh(120, 64, 3, seed=7, n_merge=300, dt=40.0, diffusion_coefficient=0.5)
    n = mesh['nreal'] + 1
    K, tr = 16, 64
    mesh = renumber_mesh(mesh, lane_order(mesh, n, tile_rows=tr))
    lm = partition_mesh(mesh['edges_face1'], mesh['edges_face2'], n, 8, 3, depth=8, align=tr)
    n_owned, n_real = lm.n_rows, lm.n_rows + lm.n_halo
    ptr, nb, edge = adjacency(lm.face1, lm.face2, n_owned, n_real)
    adv = np.asarray(mesh['face_flow'][1], dtype=np.float32)[lm.edge_global]
    plain = run(driver, tmp_path, ptr, nb, edge, adv, n_owned, lm.n_core, n_real, K, tr, 8)
    cols_plain = np.diff(plain['tptr'])
    lim = int(np.sort(cols_plain)[-4])                              # a limit that the three heaviest windows exceed
    assert cols_plain.max() > lim
    out = run(driver, tmp_path, ptr, nb, edge, adv, n_owned, lm.n_core, n_real, K, tr, 8, limits=(lim, 1 << 20))
    n_sq = int(out['n_sq'][0])
    trow, tptr, tcols, ptr2, col2 = out['trow'], out['tptr'], out['tcols'], out['ptr2'], out['col2']
    heavy = int(out['heavy'][0])
    assert 1 <= heavy <= 3 + 1 and len(trow) - 1 > len(plain['trow']) - 1
    assert trow[0] == 0 and trow[-1] == n_sq and (np.diff(trow) > 0).all()
    # every multiple of tr is a tile boundary: the windows the numbering was arranged in survive
    assert set(range(0, n_sq, tr)) <= set(trow.tolist())
    for t in range(len(trow) - 1):
        c0, c1 = int(trow[t]), int(trow[t + 1])
        want = np.unique(np.concatenate([np.arange(c0, c1), col2[ptr2[c0]:ptr2[c1]]]))
        got = tcols[tptr[t]:tptr[t + 1]]
        assert np.array_equal(np.sort(got), want) and np.array_equal(got[:c1 - c0], np.arange(c0, c1))
        assert len(got) <= lim or c1 - c0 == 1
    # links: between the tiles of the cut tiling (row ranges), never inside one
    tile_of = np.repeat(np.arange(len(trow) - 1), np.diff(trow))
    src, dst = out['link_src'], out['link_dst']
    assert (src != dst).all() and src.max() < len(trow) - 1
    pairs = set()
    for c in range(n_sq):
        for j in range(ptr[c], ptr[c + 1]):
            m = nb[j]
            if 0 <= m < n_sq and tile_of[m] != tile_of[c]:
                pairs.add((int(tile_of[c]), int(tile_of[m])))
    assert pairs == set(zip(src.tolist(), dst.tolist()))


# ---- the plan of the one-launch solver (k_small_jacobi): parts, halo layers, exchange lists --------------------------------------
def small_plan(driver, tmp_path, ptr, nb, n, threads, rpt_max, parts, depth, max_parts):
    fin, fout = str(tmp_path / 'sp_in.bin'), str(tmp_path / 'sp_out.bin')
    write_bag(fin, {'small_params': np.array([n, threads, rpt_max, parts, depth, max_parts]), 'ptr': ptr, 'nb': nb})
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=1', UBSAN_OPTIONS='print_stacktrace=1')
    res = subprocess.run([driver, fin, fout], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0 and res.stderr == '', (res.returncode, res.stderr[-3000:])
    return read_bag(fout)


def jacobi_through_plan(out, ptr, nb, w, bh, x0, sweeps):
    """The kernel's iteration restated with the plan's tables: every part relaxes all its rows (own + halo) in its own column, sums
    in the plan's neighbour order, and the halo rows are refreshed from their owners every `depth` sweeps.  Returns the global x."""
    P, rpt, depth, threads, S, R = (int(v) for v in out['dims'])
    cap = rpt * threads
    rows = out['rows'].reshape(P, cap)
    recs = out['recs'].reshape(P, 8, cap)
    offs = out['offs'].astype(np.int64).astype(np.uint32).reshape(P, 4, cap)
    gid = np.where(rows >= 0, rows & 0x0fffffff, 0)
    kind = np.where(rows >= 0, rows >> 28, 0)
    col = np.zeros((P, cap))
    for p in range(P):
        col[p] = np.where(rows[p] >= 0, x0[gid[p]], 0.0)
    pos = np.empty((P, 8, cap), dtype=np.int64)
    for q in range(8):
        half = offs[:, q // 2, :]
        pos[:, q, :] = ((half >> 16) if (q & 1) else (half & 0xffff)).astype(np.int64) // 8
    wt = np.where(recs >= 0, w[np.maximum(recs, 0)], 0.0)                    # weight of record j (zero for empty slots)
    ro = kind == 3
    wt[:, 0, :] = np.where(ro, 1.0, wt[:, 0, :])                            # read-only rows copy themselves
    b = np.where((rows >= 0) & ~ro, bh[gid], 0.0)
    since = 0
    for _ in range(sweeps):
        new = np.zeros_like(col)
        for p in range(P):
            s = np.zeros(cap)
            for q in range(8):
                s = s + wt[p, q] * col[p][pos[p, q]]
            new[p] = b[p] + s
        col = new
        since += 1
        if P > 1 and since == depth:
            pub = np.zeros((P, S))
            for p in range(P):
                c = int(out['send_cnt'][p])
                pub[p, :c] = col[p][out['send_pos'].reshape(P, S)[p, :c]]
            for p in range(P):
                c = int(out['recv_cnt'][p])
                src = out['recv_src'].reshape(P, R)[p, :c]
                col[p][out['recv_pos'].reshape(P, R)[p, :c]] = pub.reshape(-1)[src]
            since = 0
    x = np.full(len(x0), np.nan)
    for p in range(P):
        own = kind[p] == 1
        x[gid[p][own]] = col[p][own]
    return x, since


@pytest.mark.parametrize('nx,ny,n_merge,parts,depth', [(40, 20, 30, 0, 4), (109, 28, 109, 0, 4), (200, 50, 150, 0, 4), (200, 50, 150, 6, 3),
                                                      (120, 110, 300, 0, 4), (90, 45, 100, 3, 1), (64, 64, 0, 2, 2)])
def test_small_solver_plan_reproduces_the_global_jacobi_iteration_bit_for_bit(driver, tmp_path, nx, ny, n_merge, parts, depth):
    """build_small_plan under ASan / UBSan on meshes of one to seven parts: every row owned once, every slot a valid local
    position, and -- the property the halo layers exist for -- the partitioned iteration with an exchange every `depth` sweeps
    gives the SAME BITS as the global Jacobi iteration summed in the plan's neighbour order."""
    import clearwater_riverine_amd as cw
    mesh = cw.synthetic.make_mesh(nx, ny, 2, seed=nx + ny, n_merge=n_merge, n_merge4=n_merge // 3, n_dry=2, dt=40.0, diffusion_coefficient=0.5)
    n = mesh['nreal'] + 1
    ptr, nb, edge = adjacency(mesh['edges_face1'], mesh['edges_face2'], n, n)
    out = small_plan(driver, tmp_path, ptr, nb, n, 1024, 4, parts, depth, 8)
    assert out['ok'][0] == 1
    P, rpt, D, threads, S, R = (int(v) for v in out['dims'])
    cap = rpt * threads
    assert (P == 1) == (n <= 4096 and parts in (0, 1)) and (parts == 0 or P == parts) and D == (0 if P == 1 else depth)
    rows = out['rows'].reshape(P, cap)
    kind = np.where(rows >= 0, rows >> 28, 0)
    gid = rows & 0x0fffffff
    owned = np.sort(np.concatenate([gid[p][kind[p] == 1] for p in range(P)]))
    assert np.array_equal(owned, np.arange(n))                              # every row owned exactly once
    for p in range(P):
        nl = int(out['n_local'][p])
        assert np.all(rows[p, :nl] >= 0) and np.all(rows[p, nl:] == -1)
        assert len(np.unique(gid[p, :nl])) == nl                            # no row twice in a part
        recs = out['recs'].reshape(P, 8, cap)[p]
        for q in range(8):
            live = recs[q] >= 0
            assert not np.any(live[nl:]) and not np.any(live & (kind[p] == 3))
            # the record belongs to the row, and the offset points at the local position of the record's neighbour
            r = np.nonzero(live)[0]
            owner_row = np.searchsorted(ptr, recs[q][r], side='right') - 1
            assert np.array_equal(owner_row, gid[p][r])
            half = out['offs'].astype(np.int64).astype(np.uint32).reshape(P, 4, cap)[p, q // 2]
            lp = (((half >> 16) if (q & 1) else (half & 0xffff)).astype(np.int64) // 8)[r]
            assert np.all(lp < nl) and np.array_equal(gid[p][lp], nb[recs[q][r]])
    # the iteration: random weights of an M-matrix-like J (row sums < 1), random right-hand side
    rng = np.random.default_rng(5)
    w = rng.uniform(0.05, 0.2, len(nb)) * (nb >= 0)
    bh = rng.uniform(0.5, 2.0, n)
    x0 = rng.uniform(1.0, 5.0, n)
    sweeps = 3 * max(D, 1) + (0 if P == 1 else 0)
    got, since = jacobi_through_plan(out, ptr, nb, w, bh, x0, sweeps)
    assert P == 1 or since == 0
    # global Jacobi with each row's neighbours summed in the order of the part that owns it (ascending local position there)
    x = x0.copy()
    order_of_row = {}
    for p in range(P):
        recs = out['recs'].reshape(P, 8, cap)[p]
        for i in np.nonzero(kind[p] == 1)[0]:
            order_of_row[int(gid[p][i])] = [int(recs[q][i]) for q in range(8) if recs[q][i] >= 0]
    for _ in range(sweeps):
        new = np.empty(n)
        for c in range(n):
            s = 0.0
            for j in order_of_row[c]:
                s = s + w[j] * x[nb[j]]
            new[c] = bh[c] + s
        x = new
    assert np.array_equal(got, x)


def check_ell(out, K, rpw, rows_cap):
    """host::build_ell against its statement: slice s of a tile = its rows [s rpw, (s + 1) rpw), padded to the slice's longest row, entry-major;
    every CSR entry appears exactly once at pos[q] with its position code; padding carries the row's own cell; tiles start at even entries."""
    ptr2, trow, loc2 = out['ptr2'], out['trow'], out['loc2']
    eptr, sl, pos, loc = out['ell_eptr'], out['ell_sl'], out['ell_pos'], out['ell_loc']
    nsl, cap, total = (int(v) for v in out['ell_dims'])
    nt = len(trow) - 1
    assert nsl == rows_cap // rpw and len(eptr) == nt + 1 and len(sl) == nt * (nsl + 1) and total == len(loc) == eptr[nt]
    seen = np.zeros(total, dtype=bool)
    worst = 0
    for t in range(nt):
        c0, c1 = int(trow[t]), int(trow[t + 1])
        offs = sl[t * (nsl + 1):(t + 1) * (nsl + 1)]
        assert offs[0] == 0 and eptr[t] % 2 == 0
        for s_ in range(nsl):
            r0, r1 = c0 + s_ * rpw, min(c1, c0 + (s_ + 1) * rpw)
            lens = [int(ptr2[c + 1] - ptr2[c]) for c in range(r0, r1)]
            L = max(lens) if lens else 0
            assert offs[s_ + 1] - offs[s_] == L * rpw, (t, s_)
            for k in range(L):
                for r in range(rpw):
                    idx = int(eptr[t] + offs[s_] + k * rpw + r)
                    c = r0 + r
                    if c < r1 and k < lens[r]:
                        q = int(ptr2[c]) + k
                        assert pos[q] == idx and loc[idx] == loc2[q]
                        assert not seen[idx]
                        seen[idx] = True
                    else:
                        assert loc[idx] == ((c - c0) * K if c < r1 else 0)
        used = int(offs[nsl])
        assert eptr[t + 1] - eptr[t] == used + (used & 1)
        worst = max(worst, used + (used & 1))
    assert cap == max(worst, 2) and int(seen.sum()) == int(ptr2[trow[nt]]) and (pos[:int(ptr2[trow[nt]])] >= 0).all()


@pytest.mark.parametrize('K,G,ut,seed', [(16, 4, 1, 1), (1, 1, 1, 2), (8, 2, 1, 3), (4, 2, 2, 4), (32, 8, 2, 5)])
def test_wave_sliced_entry_layout_of_the_tiled_pass(driver, tmp_path, K, G, ut, seed):
    """Round 6: the sliced (ELLPACK-per-wave) entry layout the tiled pass reads -- under ASan / UBSan, against a numpy statement -- on meshes
    with merged 5-8-sided cells and dry cells, lane-group sizes G = 1 ... 8 (rows per wave 64 ... 8), one and two row sets per lane group."""
    mesh = cw.synthetic.make_mesh(37, 23, 2, seed=seed, n_merge=60, n_merge4=12, n_dry=3)
    n = mesh['nreal'] + 1
    f1, f2 = np.asarray(mesh['edges_face1']), np.asarray(mesh['edges_face2'])
    ptr, nb, edge = adjacency(f1, f2, n, n)
    rpw, R = 64 // G, 256 // G
    tr = R * ut if ut == 1 else R + R // 2                  # (two row sets: the second one part full)
    out = run(driver, tmp_path, ptr, nb, edge, np.asarray(mesh['face_flow'])[0], n, n, n, K, tr, 64, limits=(1 << 30, 1 << 30, rpw, ut * R))
    assert out['tiled'][0] == 1 and out['ell_ok'][0] == 1
    check_ell(out, K, rpw, ut * R)
