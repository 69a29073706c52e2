"""One process per GPU: the partitioned transport engine and its control plane.

Data path: every rank owns a contiguous range of real cells (partition.py); operator inputs are
completed by a neighbour halo exchange (RCCL ncclSend/ncclRecv grouped on the engine's stream, over the
direct xGMI links) and inner products by ncclAllReduce -- both issued from the C++ solver loop
(csrc/cwr_engine.hip), never from Python.  Control plane (rendezvous, broadcast of the RCCL unique id,
barriers, gathering results): torch.distributed, any backend.
"""
from __future__ import annotations

import os

import numpy as np

from .engine import TransportEngine, tile_rows
from .model import face_to_face_distance, change_in_time
from .ordering import balance_windows, hilbert_order, lane_order
from .partition import LocalMesh, partition_mesh, slice_fields


def env_rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def broadcast_bytes(payload: bytes | None, nbytes: int, src: int = 0) -> bytes:
    """Broadcast a small byte string over the default torch.distributed group (CPU or GPU backend)."""
    import torch
    import torch.distributed as dist
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    if dist.get_rank() == src:
        t.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
    dist.broadcast(t, src=src)
    return bytes(t.cpu().numpy().tobytes())


def auto_halo_depth(n_real_cells: int, world: int) -> int:
    """Halo depth for a strong-scaling run: the J^2 passes between two neighbour exchanges are depth / 2, so a deeper halo
    trades exchanges (latency-bound: pack, grouped send/recv, unpack) for replayed rows.  A compact range of m cells has a
    perimeter of about 4 sqrt(m) cells per layer; keep the replayed layers near 12-14 % of m:
    depth = 0.04 sqrt(m), even, within [8, 16].  1 M cells: 16 at 2 and 4 ranks, 14 at 8."""
    if world <= 1:
        return 1
    d = int(0.04 * np.sqrt(n_real_cells / world))
    d -= d & 1
    return int(min(16, max(8, d)))


def shared_hilbert_order(mesh: dict, n: int, rank: int, world: int, K: int = 16) -> np.ndarray:
    """order[new id] = reference id along the Hilbert curve: computed by rank 0 and broadcast when a torch.distributed
    group is up (the sort is the only global O(n log n) step of the set-up), computed locally otherwise.
    A failure on rank 0 (out of memory in the sparse product, library not built, bad mesh) reaches every rank: a status
    word travels first, so the others raise instead of waiting in the broadcast."""
    group_up = False
    if world > 1:
        try:
            import torch
            import torch.distributed as dist
            group_up = dist.is_available() and dist.is_initialized() and dist.get_world_size() == world
        except ImportError:                                   # (no torch on ANY rank of this job: all compute locally)
            group_up = False
    if not group_up:
        return _curve_order(mesh, n, K, world)
    if dist.get_rank() != rank:
        raise ValueError(f'shared_hilbert_order: rank argument {rank} is not this process\'s rank {dist.get_rank()}')
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    status = torch.zeros(1, dtype=torch.int64, device=dev)
    order, err = None, None
    if rank == 0:
        try:
            order = np.ascontiguousarray(_curve_order(mesh, n, K, world), dtype=np.int64)
            if order.shape != (n,):
                raise ValueError(f'curve order has shape {order.shape}, expected {(n,)}')
            status[0] = 1
        except Exception as ex:                               # noqa: BLE001 -- whatever it was, the other ranks must hear of it
            err = ex
    dist.broadcast(status, src=0)
    if int(status.item()) != 1:
        if err is not None:
            raise err
        raise RuntimeError('shared_hilbert_order: rank 0 failed to compute the cell order (see its traceback)')
    t = torch.empty(n, dtype=torch.int64, device=dev)
    if rank == 0:
        t.copy_(torch.from_numpy(order))
    dist.broadcast(t, src=0)
    return t.cpu().numpy()


def curve_kind(n: int, K: int, world: int = 1) -> str:
    """'lanes' or 'hilbert': which numbering _curve_order (and the facade, model.py) builds for n real cells, K constituents and
    `world` ranks.  Lanes along the flow are for engines that will CHAIN their tiles; the ENGINE says from how many rows it does
    (engine.chain_min_rows = cwr_chain_min_rows: 1.75 tiles per block of its persistent grid -- about 115 k cells per engine at K = 16,
    460 k at K = 1 --, CWR_CHAIN_MIN_TILES / CWR_TCL_GRID included: one threshold for the numbering and the engine).
    Below that the passes ping-pong between two vectors, which lose on anisotropic tiles, and the isotropic Hilbert curve is kept -- on
    one GPU too since round 4 (profiles/r04_f_small_engines.txt: 0.699 vs 0.754 ms per step at 119 k cells x 16)."""
    from .engine import chain_min_rows
    want = os.environ.get('CWR_TILE_ORDER', 'auto')
    lanes = want == 'lanes' or (want == 'auto' and not os.environ.get('CWR_NO_CHAINS') and n // max(1, world) >= chain_min_rows(K))
    return 'lanes' if lanes else 'hilbert'


def _curve_order(mesh: dict, n: int, K: int, world: int = 1) -> np.ndarray:
    """Lane-major or Hilbert order (see below), the cells of every tile-sized window sorted by their J^2 row length
    (ordering.balance_windows; the tile size of the engine's sweep kernel depends on K: cwr_tile_rows).  CWR_NO_BALANCE=1: the
    plain curve (A/B)."""
    # Engines that will run the chained passes (tile chains along the flow: from three tiles per resident block up, i.e. about
    # 200 k cells per rank at K = 16) want lanes along the flow; the ping-pong passes of smaller ranks lose from anisotropic
    # tiles (3.62 -> 4.07 ms per step on the bench mesh) and keep the isotropic Hilbert curve, which also gives compact rank
    # ranges (fewer halo rows).  CWR_TILE_ORDER=lanes|hilbert overrides.
    tr = tile_rows(K)
    lanes = curve_kind(n, K, world) == 'lanes'
    order = lane_order(mesh, n, tile_rows=tr) if lanes else hilbert_order(mesh['face_x'], mesh['face_y'], n)
    if os.environ.get('CWR_NO_BALANCE'):
        return order
    return balance_windows(order, mesh['edges_face1'], mesh['edges_face2'], window=tile_rows(K))


class PartitionedTransport:
    """The transport engine of one rank of a domain-decomposed run.  With world == 1 it is exactly the
    single-GPU engine (no halo, no communicator)."""

    def __init__(self, mesh: dict, inputs3: np.ndarray, rank: int, world: int, device: int = 0,
                 unique_id: bytes | None = None, halo_depth: int = 1, renumber: str | None = 'hilbert',
                 standalone: bool = False, flow_window: int | None = None):
        """renumber='hilbert': work in a space-filling-curve numbering of the real cells (ordering.py); every
        array handed in or out of this class stays in the reference's numbering.
        halo_depth=0: choose the depth from the size of a rank's range (auto_halo_depth).
        standalone=True (measurement only, tools/rank_step_profile.py): build rank `rank` of `world` -- its real partition, core +
        replayed layers + read-only layer -- WITHOUT a communicator: the halo rows keep what they hold, no exchange, no all-reduce.
        What such an engine computes per step is the compute side of that rank's step."""
        n = int(np.asarray(mesh['edges_face1']).max()) + 1
        if halo_depth == 0:
            halo_depth = auto_halo_depth(n, world)
        # (round 6) a LEVEL SOURCE instead of the three (T, .) arrays (levels.py: `.read(t0, t1)` or a callable (t0, t1) -> the levels of the
        # WHOLE mesh in the reference's order): the rank streams its slices through its ring -- the feeder cuts its faces and cells out of
        # every chunk in the staging block -- and never holds more than W levels of them
        level_source = mesh.get('level_source') if 'level_source' in mesh else getattr(mesh, 'attrs', {}).get('level_source')
        if level_source is not None:
            from .levels import as_level_source
            level_source = as_level_source(level_source, len(np.asarray(mesh['dt'] if 'dt' in mesh else mesh['time_seconds'])),
                                           len(mesh['edges_face1']), len(mesh['face_x']))
            if not flow_window:
                flow_window = 16
            if 'face_flow' not in mesh and renumber == 'hilbert' and curve_kind(n, int(inputs3.shape[2]), world) == 'lanes':
                mesh = dict(mesh)                                # (the lanes follow the flow of the first levels: a performance choice only)
                mesh['face_flow'] = np.asarray(level_source.read(0, min(level_source.n_times, 8))[0], dtype=np.float32)
        ncell = len(mesh['face_x'])
        self.order = None
        f1 = np.asarray(mesh['edges_face1'])
        f2 = np.asarray(mesh['edges_face2'])
        if renumber == 'hilbert':
            # the curve order is computed ONCE (rank 0) and broadcast over the control plane; no rank materialises a
            # renumbered copy of the global mesh: only the face tables are mapped (2 E integers), every field is sliced
            # straight from the reference arrays through the composed index maps below
            self.order = shared_hilbert_order(mesh, n, rank, world, int(inputs3.shape[2]))
            inv = np.arange(ncell, dtype=np.int64)
            inv[self.order] = np.arange(n)
            f1, f2 = inv[f1], inv[f2]
        elif renumber is not None:
            raise ValueError(f'unknown renumbering {renumber!r}')
        self.n_global = n
        self.K = int(inputs3.shape[2])
        self.numbering = 'reference' if renumber is None else curve_kind(n, self.K, world)
        # (rank ranges start at multiples of the tile size when the numbering was arranged in tile-sized windows)
        align = tile_rows(self.K) if (renumber == 'hilbert' and not os.environ.get('CWR_NO_BALANCE')) else 1
        self.local: LocalMesh = partition_mesh(f1, f2, n, world, rank, depth=halo_depth, align=max(1, align))
        lm = self.local
        # reference id of every local cell (ghost ids are never renumbered)
        ref_cells = lm.cell_global if self.order is None else np.where(lm.cell_global < n, self.order[np.minimum(lm.cell_global, n - 1)], lm.cell_global)
        dist_e = mesh.get('face_to_face_dist')
        if dist_e is None:
            dist_e = face_to_face_distance(mesh)                 # (reference numbering: a face keeps its two cells)
        dt = mesh.get('dt')
        if dt is None:
            dt = change_in_time(mesh['time_seconds'] if 'time_seconds' in mesh else mesh['time'])
        fields = slice_fields(lm, mesh, np.asarray(dist_e), ref_cells) if level_source is None else {'face_to_face_dist': np.ascontiguousarray(np.asarray(dist_e)[lm.edge_global])}
        self.engine = TransportEngine(lm.face1, lm.face2, lm.n_cells, self.K, n_owned=lm.n_rows,
                                      n_halo=lm.n_halo, device=device)
        # flow_window=W: a ring of W levels on the device, refilled one level per step on the engine's flow stream (cwr_flow_window_open / _load);
        # the rank's slices stay on the host, page-locked so that the uploads are asynchronous.  Round 6: ranks of a partition too (each its
        # slices; the level's norms are all-reduced where it is loaded) -- the first levels go in once the communicator is attached, below.
        self._window = None
        self._feeder = None
        if flow_window and level_source is not None:
            from .levels import FlowWindowFeeder
            T = len(np.asarray(dt))
            self._window = max(2, min(int(flow_window), T))
            self._T = T
            self.engine.flow_window_open(T, self._window, dt, fields['face_to_face_dist'], float(mesh['diffusion_coefficient']))
            self._feeder = FlowWindowFeeder(self.engine, level_source, T, self._window, cell_cols=ref_cells, edge_idx=lm.edge_global)
        elif flow_window:
            T = len(np.asarray(dt))
            self._window = max(2, min(int(flow_window), T))
            self._fields = (fields['face_flow'], fields['edge_velocity'], fields['volume'])
            self._pinned = [a for a in self._fields if self.engine.host_register(a)]
            self._T, self._win_lo, self._win_hi = T, 0, 0
            self.engine.flow_window_open(T, self._window, dt, fields['face_to_face_dist'], float(mesh['diffusion_coefficient']))
        else:
            self.engine.load_flow_field(fields['face_flow'], fields['edge_velocity'], fields['volume'], dt,
                                        fields['face_to_face_dist'], float(mesh['diffusion_coefficient']))
        ghost_global = lm.cell_global[lm.n_real:]
        # inputs3: the dense (T, ncell, K) input array, or a provider of the slices a rank needs (synthetic.DistinctInputs: initial_rows /
        # ghost_columns / real_input_entries) -- the dense array of the bench workload is 3.3 GB per rank at 1 M cells, 13 GB at 4 M
        lazy = hasattr(inputs3, 'ghost_columns')
        self.engine.load_boundary(np.ascontiguousarray(inputs3.ghost_columns(ghost_global) if lazy else inputs3[:, ghost_global, :]))
        self.standalone = bool(standalone) and world > 1
        if self.standalone:
            # a communicator of ONE rank without peers (cwr_attach_comm): the engine takes the row layout and the launch structure of a
            # rank (n_core, replayed layers, exchange-free stretches) and never exchanges
            self.engine.attach_comm(0, 1, TransportEngine.comm_unique_id(), np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32),
                                    np.zeros(1, np.int32), np.zeros(0, np.int32), n_core=lm.n_core, exchange_every=lm.depth)
        if world > 1 and not self.standalone:
            if unique_id is None:
                raise ValueError('world > 1 needs the RCCL unique id broadcast from rank 0')
            self.engine.attach_comm(rank, world, unique_id, lm.peers, lm.send_ptr, lm.send_cells, lm.recv_ptr,
                                    lm.recv_cells, n_core=lm.n_core, exchange_every=lm.depth)
            # (attach_comm also makes every rank's ||J||_inf values the maximum over the ranks: the element-wise stopping rule
            # is then the one of the global matrix, as on one GPU -- cwr_get_jacobi_norms)
        # initial condition of the owned cells (row 0 of input_array, constituents.py:94-98)
        core_ref = ref_cells[:lm.n_core]
        self.engine.set_state(np.ascontiguousarray(inputs3.initial_rows(core_ref) if lazy else inputs3[0, core_ref, :]))
        # non-zero input_array entries on REAL cells at levels >= 1 (point sources / fixed concentrations inside the domain):
        # the reference writes them into the solved level before the mass fluxes and uses them as x_t of the next step
        # (transport.py:258-264, linalg.py:199-200); every rank loads the entries of the cells it owns
        lvs, ces, vas = [], [], []
        if lazy:
            lv, ce, va = inputs3.real_input_entries(core_ref)
            if len(lv):
                lvs.append(np.asarray(lv, np.int32)); ces.append(np.asarray(ce)); vas.append(np.asarray(va))
        for t in range(1, 0 if lazy else inputs3.shape[0]):
            if not inputs3[t, :n].any():                          # (one contiguous scan per level; the usual case: nothing)
                continue
            blk = np.asarray(inputs3[t, core_ref, :])
            ce = np.nonzero(np.any(blk != 0, axis=1))[0]
            if len(ce):
                lvs.append(np.full(len(ce), t, dtype=np.int32)); ces.append(ce); vas.append(blk[ce])
        if lvs:
            self.engine.load_real_inputs(np.concatenate(lvs), np.concatenate(ces), np.concatenate(vas))
        elif world > 1 and not self.standalone:                   # (collective for partitioned engines: also with no entries)
            self.engine.load_real_inputs(np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros((0, self.K)))
        self.fill_window(0)                                       # (windowed: the ring's first levels, behind the communicator's attachment)

    def fill_window(self, t: int):
        """Windowed flow field: make levels t .. t + W - 1 resident (levels below t are not read by step t or any later one).  Only
        enqueued; a jump back in time reloads from level t."""
        if self._window is None:
            return
        if self._feeder is not None:
            self._feeder.fill(t)
            return
        if t < self._win_lo or t >= self._win_hi:                # (first call, or a jump: nothing of the ring is of use)
            self._win_hi = t
        self._win_lo = t
        hi = min(self._T, t + self._window)
        if hi > self._win_hi:
            ff, ev, vol = self._fields
            lo = self._win_hi
            self.engine.flow_window_load(lo, ff[lo:hi], ev[lo:hi], vol[lo:hi], engine_order=True)
            self._win_hi = hi

    def step(self, t: int, **kw):
        self.fill_window(t)
        return self.engine.step(t, **kw)

    def close(self):
        """Release the staging blocks of a streamed flow field (page locks), then the engine."""
        if getattr(self, '_feeder', None) is not None:
            if self.engine._h:
                self.engine.synchronize()
            self._feeder.close()
            self._feeder = None
        self.engine.close()

    def owned_state(self) -> np.ndarray:
        return self.engine.get_state()[: self.local.n_core]

    def owned_reference_ids(self) -> np.ndarray:
        """Reference cell ids of the rows owned_state() returns."""
        ids = np.arange(self.local.lo, self.local.hi)
        return ids if self.order is None else self.order[ids]

    # ------------------------------------------------------------------ output side
    def _sum_over_ranks(self, arr: np.ndarray) -> np.ndarray:
        if self.local.world == 1:
            return arr
        import torch.distributed as dist
        parts = [None] * self.local.world
        dist.all_gather_object(parts, arr)
        out = parts[0].copy()
        for p in parts[1:]:                                       # fixed rank order: identical on every rank
            out = out + p
        return out

    def set_boundary_lines(self, lines):
        """lines: face-id arrays in the REFERENCE's face numbering, one per boundary-condition line.  Each rank
        registers the faces it holds; the kernel adds only those whose face1 it owns."""
        eg = self.local.edge_global
        local = []
        for faces in lines:
            f = np.asarray(faces, dtype=np.int64).ravel()
            pos = np.searchsorted(eg, f)
            ok = (pos < len(eg)) & (eg[np.minimum(pos, len(eg) - 1)] == f)
            local.append(pos[ok].astype(np.int32))
        self.engine.set_boundary_lines(local)

    def mass_balance(self) -> np.ndarray:
        """(n_lines, 3, K) ledger summed over the ranks."""
        return self._sum_over_ranks(self.engine.get_mass_balance())

    def domain_mass(self, t_level: int):
        mass, vol = self.engine.domain_mass(t_level)
        tot = self._sum_over_ranks(np.append(mass, vol))
        return tot[:-1], float(tot[-1])

    def gather_state(self) -> np.ndarray:
        """(n_global, K) concentrations of all real cells, in the reference's numbering, on every rank
        (control-plane all_gather)."""
        mine = self.owned_state()
        if self.local.world > 1:
            import torch.distributed as dist
            parts = [None] * self.local.world
            dist.all_gather_object(parts, mine)
            mine = np.concatenate(parts, axis=0)
        if self.order is None:
            return mine
        out = np.empty_like(mine)
        out[self.order] = mine
        return out


# ------------------------------------------------------------------ constituent groups x cell ranges (round 6)
class ConstituentSlice:
    """Columns [k0, k1) of an input provider (synthetic.DistinctInputs: initial_rows / ghost_columns / real_input_entries)."""

    def __init__(self, inputs, k0: int, k1: int):
        self._in, self.k0, self.k1 = inputs, int(k0), int(k1)
        self.shape = tuple(inputs.shape[:2]) + (self.k1 - self.k0,)

    def initial_rows(self, cells):
        return np.ascontiguousarray(self._in.initial_rows(cells)[:, self.k0:self.k1])

    def ghost_columns(self, ghost_cells):
        return np.ascontiguousarray(self._in.ghost_columns(ghost_cells)[:, :, self.k0:self.k1])

    def real_input_entries(self, cells):
        lv, ce, va = self._in.real_input_entries(cells)
        return lv, ce, np.ascontiguousarray(np.asarray(va)[:, self.k0:self.k1])


def group_layout(rank: int, world: int, k_groups: int, K: int):
    """world ranks as (world / k_groups) contiguous CELL RANGES x k_groups GROUPS OF CONSTITUENTS: rank -> (group, range, ranges, k0, k1).
    Ranges vary fastest: the ranks of one group -- the only ones that ever exchange -- are neighbours in rank order."""
    if k_groups < 1 or world % k_groups != 0 or k_groups > K:
        raise ValueError(f'k_groups = {k_groups} must divide the world ({world}) and not exceed the constituents ({K})')
    R = world // k_groups
    g, r = divmod(rank, R)
    cuts = [(K * i) // k_groups for i in range(k_groups + 1)]
    return g, r, R, cuts[g], cuts[g + 1]


class GroupedTransport:
    """N GPUs as R cell ranges x G groups of K / G constituents.  The K systems of a step share A and never talk to each other (the
    reference solves them one after the other, transport.py:231-249), so a group is a complete partitioned run of ITS constituents over R
    ranges with a communicator of its own: no new kernel and no new collective -- groups never exchange anything.  Against N ranges of all K
    constituents a rank owns N / R times the cells (longer tile lists, a smaller share of replayed halo rows, at most R - 1 instead of N - 1
    peers) and moves 1 / G of the bytes per exchanged row.  A refinement of the contiguous-range partition of SURVEY 8e, not a replacement:
    k_groups = 1 IS PartitionedTransport.  Priced in profiles/r06_rank_budget.txt (1 M cells x 16 on 8 GPUs: compute-side ceiling 2.5-2.7 x as
    8 ranges, 2.9 x as 4 ranges x 2 groups)."""

    def __init__(self, mesh: dict, inputs3, rank: int, world: int, k_groups: int = 1, device: int = 0, unique_id: bytes | None = None, **kw):
        K = int(inputs3.shape[2])
        self.group, self.range, self.ranges, self.k0, self.k1 = group_layout(rank, world, k_groups, K)
        self.k_groups, self.K_all = int(k_groups), K
        sub = ConstituentSlice(inputs3, self.k0, self.k1) if hasattr(inputs3, 'ghost_columns') else np.ascontiguousarray(inputs3[:, :, self.k0:self.k1])
        self.part = PartitionedTransport(mesh, sub, self.range, self.ranges, device=device, unique_id=unique_id, **kw)
        self.engine, self.local, self.K = self.part.engine, self.part.local, self.part.K
        self.numbering, self.order = self.part.numbering, self.part.order

    def step(self, t: int, **kw):
        return self.part.step(t, **kw)

    def fill_window(self, t: int):
        return self.part.fill_window(t)

    def owned_state(self):
        return self.part.owned_state()

    def owned_reference_ids(self):
        return self.part.owned_reference_ids()
