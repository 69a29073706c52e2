"""Domain decomposition of the transport path: contiguous cell-id ranges, one per GPU, with
optionally DEEP halos so that several Jacobi sweeps run between two neighbour exchanges.

The reference is single-process; this is the MI355X-native layout of the north
star: real cells are split by contiguous id range over the ranks of one node.  With halo depth s a rank
holds, besides its core range, the s layers of real cells around it (layer l = cells at face-distance l
from the core).  It COMPUTES the core and layers 1..s-1 and only reads layer s: after an exchange all
layers are exact; every sweep makes one more outer layer stale, and after s sweeps only the core is
exact -- which is all the neighbours need at the next exchange.  s = 1 is the classic one-exchange-per-
operator scheme.  The redundant rows replay the owner's arithmetic bit for bit (same kernel, faces of
a row in ascending global id), so results do not depend on s.

The boundary ghost cells of the reference (ids > nreal, /root/reference/src/clearwater_riverine/
io/hdf.py:268-269) are NOT halo: their values come from input_array and live with every rank that
computes the adjacent row.

Everything here is pure index logic on the host (numpy), identical on every rank, so that send and
receive lists agree without any negotiation.  Local numbering handed to the engine:
    [0, n_core)                 core rows (global id - lo)
    [n_core, n_rows)            halo layers 1..s-2 merged in ascending global id, then layer s-1
    [n_rows, n_rows + n_halo)   halo layer s (read only)
    [n_rows + n_halo, n_cells)  ghost cells, ascending global id
Local faces keep ascending global face id (the reference's last-write-wins order, linalg.py:349-351).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


def range_bounds(n_real_cells: int, world: int, align: int = 1) -> np.ndarray:
    """lo/hi of every rank: bounds[r] .. bounds[r+1].  align > 1: the inner bounds are rounded to multiples of `align` (the
    tile size of the engine's sweep kernel), so that a rank's tiles coincide with the tile-sized windows the global
    numbering was arranged in (ordering.balance_windows); ranges then differ by at most `align` cells."""
    b = (np.arange(world + 1, dtype=np.int64) * n_real_cells) // world
    if align > 1 and n_real_cells >= 2 * align * world:
        b[1:-1] = ((b[1:-1] + align // 2) // align) * align
    return b


def halo_layers(f1: np.ndarray, f2: np.ndarray, n_real_cells: int, lo: int, hi: int, depth: int):
    """Breadth-first layers of real cells around the id range [lo, hi): list of sorted id arrays."""
    real2 = f2 < n_real_cells
    a, b = f1[real2], f2[real2]                               # faces between two real cells
    inside = np.zeros(n_real_cells, dtype=bool)
    inside[lo:hi] = True
    layers = []
    for _ in range(depth):
        ia, ib = inside[a], inside[b]
        new = np.unique(np.concatenate([b[ia & ~ib], a[ib & ~ia]]))
        if len(new) == 0:
            layers.append(new)
            continue
        inside[new] = True
        layers.append(new)
    return layers


@dataclass
class LocalMesh:
    rank: int
    world: int
    lo: int
    hi: int
    depth: int
    n_core: int                       # rows this rank owns (norms, results)
    n_rows: int                       # rows it computes (core + layers 1..depth-1)
    n_halo: int                       # read-only rows (layer depth)
    n_cells: int                      # rows + halo + ghost
    cell_global: np.ndarray           # (n_cells,) global id of every local cell
    edge_global: np.ndarray           # (n_edges_local,) global face ids, ascending
    face1: np.ndarray                 # local ids
    face2: np.ndarray
    peers: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    send_ptr: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    send_cells: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))   # local core ids, per peer
    recv_ptr: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    recv_cells: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))   # local halo ids, per peer

    # names used before deep halos existed
    @property
    def n_owned(self) -> int:
        return self.n_core

    @property
    def n_real(self) -> int:
        return self.n_rows + self.n_halo

    @property
    def n_ghost(self) -> int:
        return self.n_cells - self.n_real


def partition_mesh(face1, face2, n_real_cells: int, world: int, rank: int, depth: int = 1, align: int = 1) -> LocalMesh:
    """Local mesh of `rank` for real cells [0, n_real_cells) split into `world` contiguous ranges."""
    f1 = np.asarray(face1, dtype=np.int64)
    f2 = np.asarray(face2, dtype=np.int64)
    if depth < 1:
        raise ValueError('halo depth must be >= 1')
    if f1.max(initial=-1) >= n_real_cells:
        raise ValueError('face1 must be a real cell for every face')
    bounds = range_bounds(n_real_cells, world, align)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    n_core = hi - lo
    if n_core <= 0:
        raise ValueError(f'rank {rank} owns no cells ({n_real_cells} cells over {world} ranks)')
    if world == 1:
        depth = 1
    layers = halo_layers(f1, f2, n_real_cells, lo, hi, depth)
    # computed halo rows: layers 1..s-2 merged in ascending global id (with a space-filling-curve numbering that keeps
    # the strip around the core spatially coherent, so the engine can tile it like the core), then layer s-1
    if depth > 2:
        computed_halo = np.concatenate([np.sort(np.concatenate(layers[:-2])), layers[-2]])
    elif depth == 2:
        computed_halo = layers[0]
    else:
        computed_halo = np.zeros(0, dtype=np.int64)
    last = layers[-1]
    n_rows = n_core + len(computed_halo)
    n_halo = len(last)

    # local id of every global real cell we hold (-1 elsewhere)
    lid = np.full(n_real_cells, -1, dtype=np.int64)
    lid[lo:hi] = np.arange(n_core)
    lid[computed_halo] = n_core + np.arange(len(computed_halo))
    lid[last] = n_rows + np.arange(n_halo)

    # faces: every face touching a computed row
    n_is_real = f2 < n_real_cells
    l1 = lid[f1]
    l2 = np.where(n_is_real, lid[np.minimum(f2, n_real_cells - 1)], -1)
    comp1 = (l1 >= 0) & (l1 < n_rows)
    comp2 = n_is_real & (l2 >= 0) & (l2 < n_rows)
    eg = np.nonzero(comp1 | comp2)[0]                          # ascending global face id
    p, q = f1[eg], f2[eg]
    q_real = n_is_real[eg]
    ghosts = np.unique(q[~q_real])
    cell_global = np.concatenate([np.arange(lo, hi, dtype=np.int64), computed_halo, last, ghosts])
    lf1 = lid[p]
    lf2 = np.where(q_real, lid[np.minimum(q, n_real_cells - 1)], n_rows + n_halo + np.searchsorted(ghosts, q))
    if lf1.min(initial=0) < 0 or lf2.min(initial=0) < 0:
        raise AssertionError('partition: a local face references a cell outside the halo closure')

    # receive lists: every halo cell (all layers) from its owner, ascending global id per owner
    halo_all = np.concatenate([computed_halo, last])
    owner = np.searchsorted(bounds, halo_all, side='right') - 1
    peers = np.unique(owner).astype(np.int32)
    send_ptr, send_cells, recv_ptr, recv_cells = [0], [], [0], []
    for s in peers:
        want = np.sort(halo_all[owner == s])
        recv_cells.append(lid[want])
        recv_ptr.append(recv_ptr[-1] + len(want))
        # what rank s wants from us: its halo layers intersected with our range (same BFS, seeded with ITS range)
        theirs = np.concatenate(halo_layers(f1, f2, n_real_cells, int(bounds[s]), int(bounds[s + 1]), depth))
        give = np.sort(theirs[(theirs >= lo) & (theirs < hi)])
        send_cells.append(give - lo)
        send_ptr.append(send_ptr[-1] + len(give))
    return LocalMesh(
        rank=rank, world=world, lo=lo, hi=hi, depth=depth, n_core=n_core, n_rows=n_rows, n_halo=n_halo,
        n_cells=len(cell_global), cell_global=cell_global, edge_global=eg,
        face1=lf1.astype(np.int32), face2=lf2.astype(np.int32), peers=peers,
        send_ptr=np.asarray(send_ptr, dtype=np.int32),
        send_cells=(np.concatenate(send_cells) if send_cells else np.zeros(0, np.int64)).astype(np.int32),
        recv_ptr=np.asarray(recv_ptr, dtype=np.int32),
        recv_cells=(np.concatenate(recv_cells) if recv_cells else np.zeros(0, np.int64)).astype(np.int32))


def slice_fields(local: LocalMesh, mesh: dict, dist: np.ndarray, ref_cells: np.ndarray | None = None) -> dict:
    """Per-rank slices of the flow field in local numbering (what the rank uploads to its GPU).  `mesh` is in the
    reference numbering; ref_cells[i] = reference id of local cell i (default: the partition was made in it)."""
    eg = local.edge_global
    cg = local.cell_global if ref_cells is None else ref_cells
    return {
        'face_flow': np.ascontiguousarray(np.asarray(mesh['face_flow'])[:, eg]),
        'edge_velocity': np.ascontiguousarray(np.asarray(mesh['edge_velocity'])[:, eg]),
        'volume': np.ascontiguousarray(np.asarray(mesh['volume'])[:, cg]),
        'face_to_face_dist': np.ascontiguousarray(dist[eg]),
    }
