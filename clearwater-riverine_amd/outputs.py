"""Streamed output of the transport path.

The reference keeps every constituent as a RAM-resident (T, ncell) float64 array plus three (T, nedge) flux arrays
(/root/reference/src/clearwater_riverine/constituents.py:28-48) and writes the whole xarray Dataset at the end
(io/outputs.py:11-22, ``mesh.to_zarr(..., consolidated=True)``).  At 4 M cells x 16 constituents one time level of that
model is 0.5 GB of state and 1.5 GB of fluxes, so a time series cannot be held at all.  Here every step's state leaves
the GPU through a ring of pinned host slots (cwr_output_push / _wait / _release, copies on a second HIP stream while
the next steps compute) and a writer thread appends it to a zarr-v2 directory store laid out the way xarray's
``to_zarr`` lays out the reference's Dataset: one array per constituent, dims (time, nface), one chunk per time level.
The store is plain files (JSON metadata + raw little-endian float64 chunks, no compressor), written without the zarr
package, and opens with ``xarray.open_zarr(path)``.
"""
from __future__ import annotations

import json
import os
import queue
import sys
import threading

import numpy as np


class ZarrStreamWriter:
    """Minimal zarr-v2 directory store: float64 arrays of shape (T, n) chunked (1, n), appended level by level."""

    def __init__(self, path: str, arrays: dict, n_times: int, attrs: dict | None = None):
        """arrays: name -> (n, dim_name), e.g. {'salinity': (ncell, 'nface'), 'salinity_total_mass_flux': (nedge, 'nedge')}."""
        parent = os.path.dirname(os.path.abspath(path))
        if not os.path.isdir(parent):                          # io/outputs.py:33-38
            raise FileNotFoundError(2, os.strerror(2), path)
        self.path, self.n_times = path, int(n_times)
        self.arrays = {k: (int(v[0]), str(v[1])) for k, v in arrays.items()}
        os.makedirs(path, exist_ok=True)
        meta = {'.zgroup': {'zarr_format': 2}, '.zattrs': dict(attrs or {})}
        for name, (n, dim) in self.arrays.items():
            os.makedirs(os.path.join(path, name), exist_ok=True)
            meta[f'{name}/.zarray'] = {
                'zarr_format': 2, 'shape': [self.n_times, n], 'chunks': [1, n], 'dtype': '<f8', 'compressor': None,
                'fill_value': 'NaN', 'order': 'C', 'filters': None}
            meta[f'{name}/.zattrs'] = {'_ARRAY_DIMENSIONS': ['time', dim]}
        for key, val in meta.items():
            with open(os.path.join(path, key), 'w') as fh:
                json.dump(val, fh)
        with open(os.path.join(path, '.zmetadata'), 'w') as fh:     # consolidated=True, io/outputs.py:16
            json.dump({'zarr_consolidated_format': 1, 'metadata': meta}, fh)

    def write_level(self, name: str, t: int, row: np.ndarray):
        n, _ = self.arrays[name]
        if row.shape != (n,) or row.dtype != np.float64:
            raise ValueError(f'{name}: expected a float64 row of length {n}')
        if not 0 <= t < self.n_times:
            raise IndexError(f'time level {t} outside the store')
        tmp = os.path.join(self.path, name, f'.{t}.0.part')
        with open(tmp, 'wb') as fh:
            fh.write(np.ascontiguousarray(row, dtype='<f8').data)
        os.replace(tmp, os.path.join(self.path, name, f'{t}.0'))    # a chunk is either absent (= fill value) or complete


def read_zarr_level(path: str, name: str, t: int) -> np.ndarray:
    """Read one chunk back (tests; a missing chunk is the fill value, as zarr defines)."""
    with open(os.path.join(path, name, '.zarray')) as fh:
        meta = json.load(fh)
    n = meta['shape'][1]
    f = os.path.join(path, name, f'{t}.0')
    if not os.path.exists(f):
        return np.full(n, np.nan)
    return np.fromfile(f, dtype=meta['dtype'], count=n)


class StreamedOutput:
    """Engine -> pinned ring -> writer thread -> zarr store.

    push(t) is called after the step that produced level t (t = 0 for the initial condition); it returns as soon as the
    device-side snapshot is enqueued.  close() drains the ring and joins the writer."""

    def __init__(self, engine, path: str, names, n_times: int, *, with_flux: bool = False, n_slots: int = 3,
                 attrs: dict | None = None):
        self.engine, self.names, self.with_flux = engine, list(names), bool(with_flux)
        if len(self.names) != engine.K:
            raise ValueError('one name per constituent')
        arrays = {nme: (engine.n_cells, 'nface') for nme in self.names}
        if self.with_flux:
            for nme in self.names:
                for kind in ('advection_mass_flux', 'diffusion_mass_flux', 'total_mass_flux'):
                    arrays[f'{nme}_{kind}'] = (engine.n_edges, 'nedge')
        self.writer = ZarrStreamWriter(path, arrays, n_times, attrs)
        engine.output_open(n_slots=n_slots, with_flux=self.with_flux)
        self._q: queue.Queue = queue.Queue()
        self._err: list = []
        self.levels_written = 0
        self._closed = False
        self._thread = threading.Thread(target=self._drain, name='cwr-output-writer', daemon=True)
        self._thread.start()

    def push(self, t: int):
        if self._err:
            raise self._err[0]
        slot = self.engine.output_push()           # blocks only if the writer still holds this slot (ring full)
        self._q.put((int(t), slot))

    def _drain(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            t, slot = item
            try:
                state, flux = self.engine.output_wait(slot)          # GIL released while the copy lands
                for k, nme in enumerate(self.names):
                    self.writer.write_level(nme, t, state[k])
                    if flux is not None and t >= 1:                  # fluxes of the step t-1 -> t live at row t-1 (transport.py:267-273)
                        for q, kind in enumerate(('advection_mass_flux', 'diffusion_mass_flux', 'total_mass_flux')):
                            self.writer.write_level(f'{nme}_{kind}', t - 1, flux[q, k])
                self.levels_written += 1
            except Exception as exc:                                  # surfaced by the next push() / close()
                self._err.append(exc)
            finally:
                try:
                    self.engine.output_release(slot)
                except Exception as exc:
                    self._err.append(exc)

    def close(self):
        """Drain the ring, join the writer, close the engine's ring.  Idempotent.  During interpreter finalisation (a facade
        collected after the atexit hook of model.py has run, or a caller's own late finalizer) nothing is done: the daemon
        writer thread no longer runs then -- a join would never return -- and the engine ignores late calls by itself."""
        if self._closed or sys.is_finalizing():
            return
        self._closed = True
        self._q.put(None)
        self._thread.join()
        self.engine.output_close()
        if self._err:
            raise self._err[0]
