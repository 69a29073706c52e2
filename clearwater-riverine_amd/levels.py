"""Time levels of the flow field from file to the engine's ring ("double-buffered H2D per step").

The reference reads the whole `datetime_range` of a HEC-RAS HDF into RAM (/root/reference/src/clearwater_riverine/io/hdf.py:149-191,
:275-310) and derives the coefficients of every level at once (utilities.py:513-541); its own fixture has 10 801 stamps, and a 1 M-cell
model of that length is 220 GB of host arrays.  Here the engine keeps a ring of W levels on the device (cwr_flow_window_open / _load) and

  * a LEVEL SOURCE hands out levels on demand: `source.read(t0, t1) -> (face_flow, edge_velocity, volume)`, float32,
    (t1 - t0, n_edges) x 2 and (t1 - t0, n_cells), in the reference's face / cell order.  `HdfLevelSource` reads h5py hyperslabs by
    level (same datasets and time window as io/hdf.py:275-310); `ArrayLevelSource` slices arrays; any callable `(t0, t1) -> triple` is
    wrapped by `as_level_source`;
  * the FEEDER pulls W / 2 levels at a time into two page-locked staging blocks (one host ring of W levels: a level's block is free
    again once the step that reads it as its level t + 1 has returned, which is before its slot in the device ring is loaded again),
    permutes the volumes into the engine's cell order there, and hands the block to the engine (asynchronous uploads on the engine's
    flow stream).  The boundary values of the same levels travel with them (cwr_boundary_window_load).

Host memory is O(W) levels whatever the length of the run.  Results are those of the resident run bit for bit (the ring and the
derivation per level are the engine's; tests/test_gpu_window.py).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

_BASE = 'Results/Unsteady/Output/Output Blocks/Base Output/Unsteady Time Series'


class ArrayLevelSource:
    """Levels out of arrays that are in memory anyway (or memory-mapped: np.load(mmap_mode='r'))."""

    def __init__(self, face_flow, edge_velocity, volume):
        self.face_flow, self.edge_velocity, self.volume = face_flow, edge_velocity, volume
        self.n_times = int(face_flow.shape[0])
        self.n_edges = int(face_flow.shape[1])
        self.n_cells = int(volume.shape[1])

    def read(self, t0: int, t1: int):
        return self.face_flow[t0:t1], self.edge_velocity[t0:t1], self.volume[t0:t1]

    def close(self):
        pass


class CallableLevelSource:
    """Any callable (t0, t1) -> (face_flow, edge_velocity, volume) of levels [t0, t1)."""

    def __init__(self, fn: Callable, n_times: int, n_edges: int, n_cells: int):
        self.fn, self.n_times, self.n_edges, self.n_cells = fn, int(n_times), int(n_edges), int(n_cells)

    def read(self, t0: int, t1: int):
        return self.fn(int(t0), int(t1))

    def close(self):
        pass


class HdfLevelSource:
    """Levels [first, first + n_times) of a HEC-RAS 2D result file, read by hyperslab (io/hdf.py:275-310 names the datasets:
    'Face Flow', 'Face Velocity', 'Cell Volume' under the flow area's Unsteady Time Series group; io/hdf.py:149-191 the window).
    The file is opened at the first read and stays open (h5py keeps one chunk cache per dataset, ~1 MB each)."""

    def __init__(self, file_path: str, area: str, first: int, n_times: int, n_edges: int, n_cells: int):
        self.file_path, self.area, self.first = file_path, area, int(first)
        self.n_times, self.n_edges, self.n_cells = int(n_times), int(n_edges), int(n_cells)
        self._f = None
        self._ds = None
        self.levels_read = 0                                      # (tests: every level crosses exactly once in a forward run)
        self.largest_read = 0

    def _open(self):
        if self._f is None:
            import h5py
            self._f = h5py.File(self.file_path, 'r')
            res = self._f[f'{_BASE}/2D Flow Areas/{self.area}']
            self._ds = (res['Face Flow'], res['Face Velocity'], res['Cell Volume'])

    def read(self, t0: int, t1: int):
        if not (0 <= t0 <= t1 <= self.n_times):
            raise IndexError(f'levels [{t0}, {t1}) outside the {self.n_times} levels of the window')
        self._open()
        a, b = self.first + t0, self.first + t1
        self.levels_read += t1 - t0
        self.largest_read = max(self.largest_read, t1 - t0)
        return tuple(np.asarray(d[a:b], dtype=np.float32) for d in self._ds)

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = self._ds = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def as_level_source(obj, n_times: int, n_edges: int, n_cells: int):
    if hasattr(obj, 'read'):
        return obj
    if callable(obj):
        return CallableLevelSource(obj, n_times, n_edges, n_cells)
    raise TypeError('a level source has .read(t0, t1) or is a callable (t0, t1) -> (face_flow, edge_velocity, volume)')


def _page_aligned(shape, dtype) -> np.ndarray:
    """A zeroed array on pages of its own (hipHostRegister wants whole pages)."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.zeros(n + 8192, dtype=np.uint8)
    off = (-raw.ctypes.data) % 4096
    return raw[off:off + n].view(dtype).reshape(shape)


class FlowWindowFeeder:
    """File -> staging -> ring.  One per engine (a rank of a partition passes the gathers that cut its slices out of a level).

    engine        : TransportEngine with an open flow window of W levels (flow_window_open)
    source        : level source (reference face / cell order, the WHOLE mesh)
    T, W          : levels of the run, levels of the ring
    cell_cols     : engine row -> source cell (the volumes' column gather: engine._cols of a renumbered engine, a rank's cells); None = identity
    edge_idx      : engine face -> source face (a rank's faces); None = identity
    boundary      : None, or callable (t0, t1) -> (t1 - t0, n_ghost, K) float64 boundary values of those levels (engine ghost order)
    chunk         : levels per read (default W // 2: two staging blocks)
    on_levels     : None, or callable (t0, t1, face_flow, edge_velocity, volume) called with every chunk as read (e.g. to keep the few
                    boundary-face flows a global mass balance needs)
    """

    def __init__(self, engine, source, T: int, W: int, *, cell_cols=None, edge_idx=None, boundary: Optional[Callable] = None,
                 chunk: Optional[int] = None, pin: bool = True, on_levels: Optional[Callable] = None):
        self.engine, self.source = engine, source
        self.T, self.W = int(T), max(2, min(int(W), int(T)))
        self.C = max(1, min(int(chunk) if chunk else self.W // 2, self.W))
        self.H = -(-self.W // self.C) * self.C                   # host ring: whole chunks, >= W levels
        self.cell_cols = None if cell_cols is None else np.ascontiguousarray(cell_cols, dtype=np.int64)
        self.edge_idx = None if edge_idx is None else np.ascontiguousarray(edge_idx, dtype=np.int64)
        E, nc = engine.n_edges, engine.n_cells
        self._ff = _page_aligned((self.H, E), np.float32)
        self._ev = _page_aligned((self.H, E), np.float32)
        self._vol = _page_aligned((self.H, nc), np.float32)
        self.boundary = boundary
        self.on_levels = on_levels                                # callable(a, b, face_flow, edge_velocity, volume) on every chunk as the source returned it
        self._bc = None
        if boundary is not None:
            self._bc = _page_aligned((self.H, max(1, engine.n_ghost), engine.K), np.float64)
        self._pinned = []
        if pin:
            for a in (self._ff, self._ev, self._vol, self._bc):
                if a is not None and a.nbytes >= (1 << 16) and engine.host_register(a):
                    self._pinned.append(a)
        self.hi = 0                                               # levels [.., hi) have been handed to the engine
        self.lo = 0
        self.staged_bytes = sum(a.nbytes for a in (self._ff, self._ev, self._vol, self._bc) if a is not None)

    def _stage(self, a: int, b: int):
        """Levels [a, b) (one contiguous stretch of the host ring) from the source into the staging blocks, then to the engine."""
        ff, ev, vol = self.source.read(a, b)
        if self.on_levels is not None:
            self.on_levels(a, b, ff, ev, vol)
        s0 = a % self.H
        s1 = s0 + (b - a)
        if self.edge_idx is None:
            self._ff[s0:s1] = ff
            self._ev[s0:s1] = ev
        else:
            np.take(ff, self.edge_idx, axis=1, out=self._ff[s0:s1])
            np.take(ev, self.edge_idx, axis=1, out=self._ev[s0:s1])
        if self.cell_cols is None:
            self._vol[s0:s1] = vol
        else:
            np.take(vol, self.cell_cols, axis=1, out=self._vol[s0:s1])
        self.engine.flow_window_load(a, self._ff[s0:s1], self._ev[s0:s1], self._vol[s0:s1], engine_order=True)
        if self._bc is not None:
            self._bc[s0:s1] = self.boundary(a, b)
            self.engine.boundary_window_load(a, self._bc[s0:s1])

    def fill(self, t: int):
        """Before step t: every level up to t + W - 1 that fits a whole chunk -- or that step t itself needs -- goes to the engine.
        Only enqueued (the engine uploads on its flow stream beside the steps).  A jump (t outside what the ring holds) restarts at t."""
        if t < self.lo or t > self.hi:
            self.engine.synchronize()                             # (the staging blocks are about to be reused out of turn)
            self.hi = t
        self.lo = t
        top = min(self.T, t + self.W)                             # the ring may hold levels [t, top)
        while self.hi < top:
            want = min(self.hi + self.C - self.hi % self.C, top, self.T)      # to the end of the chunk `hi` lies in
            whole = (want - self.hi == self.C - self.hi % self.C) or want == self.T
            if not whole and self.hi >= min(self.T, t + 2):
                break                                             # a partial chunk that step t does not need yet: wait for the rest to fit
            self._stage(self.hi, want)
            self.hi = want

    def close(self):
        for a in self._pinned:
            try:
                self.engine.host_unregister(a)
            except Exception:
                pass
        self._pinned = []
        self.source.close()
