"""Domain decomposition of the transport path: contiguous cell-id ranges, one per GPU.

The reference is single-process; this is the MI355X-native layout of the north
star: real cells are split by contiguous id range over the ranks of one node, a face belongs to every
rank that owns one of its two cells, the real cells a rank reads but does not own form its halo, and
the boundary ghost cells of the reference (ids > nreal, /root/reference/src/clearwater_riverine/
io/hdf.py:268-269) are NOT halo: their values come from input_array and live with the owner of face1.

Everything here is pure index logic on the host (numpy), identical on every rank, so that send and
receive lists agree without any negotiation.  Local numbering handed to the engine:
    [0, n_owned) owned real cells (global id - lo)
    [n_owned, n_owned + n_halo) halo cells ordered by (owner rank, global id)
    [n_owned + n_halo, n_cells_local) ghost cells ordered by global id
Local faces keep ascending global face id (the reference's last-write-wins order, linalg.py:349-351).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


def range_bounds(n_real_cells: int, world: int) -> np.ndarray:
    """lo/hi of every rank: bounds[r] .. bounds[r+1]."""
    return (np.arange(world + 1, dtype=np.int64) * n_real_cells) // world


@dataclass
class LocalMesh:
    rank: int
    world: int
    lo: int
    hi: int
    n_owned: int
    n_halo: int
    n_cells: int                      # owned + halo + ghost
    cell_global: np.ndarray           # (n_cells,) global id of every local cell
    edge_global: np.ndarray           # (n_edges_local,) global face ids, ascending
    face1: np.ndarray                 # local ids
    face2: np.ndarray
    peers: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    send_ptr: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    send_cells: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))   # local owned ids
    recv_ptr: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))     # offsets into the halo block

    @property
    def n_real(self) -> int:
        return self.n_owned + self.n_halo

    @property
    def n_ghost(self) -> int:
        return self.n_cells - self.n_real


def partition_mesh(face1, face2, n_real_cells: int, world: int, rank: int) -> LocalMesh:
    """Local mesh of `rank` for real cells [0, n_real_cells) split into `world` contiguous ranges."""
    f1 = np.asarray(face1, dtype=np.int64)
    f2 = np.asarray(face2, dtype=np.int64)
    if f1.max(initial=-1) >= n_real_cells:
        raise ValueError('face1 must be a real cell for every face')
    bounds = range_bounds(n_real_cells, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    n_owned = hi - lo
    if n_owned <= 0:
        raise ValueError(f'rank {rank} owns no cells ({n_real_cells} cells over {world} ranks)')
    n_is_real = f2 < n_real_cells
    own1 = (f1 >= lo) & (f1 < hi)
    own2 = n_is_real & (f2 >= lo) & (f2 < hi)
    local = own1 | own2
    eg = np.nonzero(local)[0]                                   # ascending global face id
    p, q = f1[eg], f2[eg]
    q_real = n_is_real[eg]
    # halo: real cells of local faces outside [lo, hi)
    ends = np.concatenate([p, q[q_real]])
    halo = np.unique(ends[(ends < lo) | (ends >= hi)])          # sorted by global id == (owner, id) order
    ghosts = np.unique(q[~q_real])
    n_halo = len(halo)
    cell_global = np.concatenate([np.arange(lo, hi, dtype=np.int64), halo, ghosts])

    def to_local(g):
        out = np.empty(len(g), dtype=np.int64)
        own = (g >= lo) & (g < hi)
        out[own] = g[own] - lo
        gh = g >= n_real_cells
        out[gh] = n_owned + n_halo + np.searchsorted(ghosts, g[gh])
        hl = ~own & ~gh
        out[hl] = n_owned + np.searchsorted(halo, g[hl])
        return out

    lf1 = to_local(p).astype(np.int32)
    lf2 = to_local(q).astype(np.int32)

    # receive lists: the halo block is already grouped by owner
    owner_of_halo = np.searchsorted(bounds, halo, side='right') - 1
    peers_recv = np.unique(owner_of_halo)
    # send lists: owned cells adjacent to a cell of rank s, ascending global id (== s's halo order)
    other = np.concatenate([q[own1[eg] & q_real], p[own2[eg]]])        # far end of faces whose near end is owned
    mine = np.concatenate([p[own1[eg] & q_real], q[own2[eg]]])
    far_out = (other < lo) | (other >= hi)
    other, mine = other[far_out], mine[far_out]
    owner_other = np.searchsorted(bounds, other, side='right') - 1
    peers_send = np.unique(owner_other)
    peers = np.union1d(peers_recv, peers_send).astype(np.int32)
    send_ptr = [0]
    send_cells = []
    recv_ptr = [0]
    for s in peers:
        cells = np.unique(mine[owner_other == s])
        send_cells.append(cells - lo)
        send_ptr.append(send_ptr[-1] + len(cells))
        recv_ptr.append(recv_ptr[-1] + int(np.count_nonzero(owner_of_halo == s)))
    return LocalMesh(
        rank=rank, world=world, lo=lo, hi=hi, n_owned=n_owned, n_halo=n_halo, n_cells=len(cell_global),
        cell_global=cell_global, edge_global=eg, face1=lf1, face2=lf2, peers=peers,
        send_ptr=np.asarray(send_ptr, dtype=np.int32),
        send_cells=(np.concatenate(send_cells) if send_cells else np.zeros(0, np.int64)).astype(np.int32),
        recv_ptr=np.asarray(recv_ptr, dtype=np.int32))


def slice_fields(local: LocalMesh, mesh: dict, dist: np.ndarray) -> dict:
    """Per-rank slices of the flow field in local numbering (what the rank uploads to its GPU)."""
    eg = local.edge_global
    cg = local.cell_global
    return {
        'face_flow': np.ascontiguousarray(np.asarray(mesh['face_flow'])[:, eg]),
        'edge_velocity': np.ascontiguousarray(np.asarray(mesh['edge_velocity'])[:, eg]),
        'volume': np.ascontiguousarray(np.asarray(mesh['volume'])[:, cg]),
        'face_to_face_dist': np.ascontiguousarray(dist[eg]),
    }
