"""Host-side mirror of the reference's model facade for the transport path.

``ClearwaterRiverine`` here keeps the constructor keywords, attributes and ``update()`` semantics of
/root/reference/src/clearwater_riverine/transport.py:68-276 for the per-step transport path, with the
LHS/RHS assembly + scipy.sparse solve replaced by the HIP engine (``engine.TransportEngine``).

The Dataset surface is a plain mapping of numpy arrays keyed by the reference's variable names
(variables.py:1-37) because xarray is not installed in this image; when the reference runs with its
own xarray mesh, INTEGRATION.md shows the few lines that hand ``mesh[var].values`` to the same engine.
Out of scope here: plotting, zarr/netCDF output, unit conversion, YAML config.
"""
from __future__ import annotations

import sys
from typing import Any, Dict, Optional

import numpy as np

from .mass_balance import assemble, boundary_lines, volume_columns
from .outputs import StreamedOutput, ZarrStreamWriter
from .engine import TransportEngine, StepResult, tile_rows
from .ordering import balance_windows, hilbert_order, lane_order

# variables.py names used on the path
EDGES_FACE1 = 'edges_face1'
EDGES_FACE2 = 'edges_face2'
NUMBER_OF_REAL_CELLS = 'nreal'
VOLUME = 'volume'
EDGE_VELOCITY = 'edge_velocity'
CHANGE_IN_TIME = 'dt'
FLOW_ACROSS_FACE = 'face_flow'
ADVECTION_COEFFICIENT = 'advection_coeff'
FACE_TO_FACE_DISTANCE = 'face_to_face_dist'
COEFFICIENT_TO_DIFFUSION_TERM = 'coeff_to_diffusion'


class Mesh(dict):
    """Minimal stand-in for the reference's xr.Dataset: variables by name + ``attrs`` +
    the attribute shortcuts the transport path uses (``mesh.nreal``, ``mesh.diffusion_coefficient``)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.attrs: Dict[str, Any] = {}

    def __getattr__(self, name):
        attrs = self.__dict__.get('attrs', {})
        if name in attrs:
            return attrs[name]
        if name in self:
            return self[name]
        raise AttributeError(name)


def face_to_face_distance(mesh) -> np.ndarray:
    """utilities.py:261-278 (_calc_distances_cell_centroids), float64."""
    f1 = np.asarray(mesh[EDGES_FACE1])
    f2 = np.asarray(mesh[EDGES_FACE2])
    fx = np.asarray(mesh['face_x'], dtype=np.float64)
    fy = np.asarray(mesh['face_y'], dtype=np.float64)
    return np.sqrt((fx[f1] - fx[f2]) ** 2 + (fy[f1] - fy[f2]) ** 2)


def change_in_time(time) -> np.ndarray:
    """utilities.py:537-541: seconds between stamps with a trailing NaN.  ``time`` is datetime64 or seconds."""
    t = np.asarray(time)
    if np.issubdtype(t.dtype, np.datetime64):
        d = np.ediff1d(t) / np.timedelta64(1, 's')
    else:
        d = np.ediff1d(t.astype(np.float64))
    return np.append(d.astype(np.float64), np.nan)


def _page_block(shape, fill: float) -> np.ndarray:
    """float64 array on pages of its own (anonymous mmap, whole pages): it can be page-locked (hipHostRegister pins whole pages)
    without pulling unrelated heap objects that share its first or last page into the locked range -- a later copy from such an
    object fails with 'invalid argument'."""
    import mmap
    count = int(np.prod(shape))
    nbytes = max(1, count) * 8
    buf = mmap.mmap(-1, (nbytes + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE)
    arr = np.frombuffer(buf, dtype=np.float64, count=count).reshape(shape)
    arr[...] = fill
    return arr


class SparseInputArray:
    """`input_array` (T, ncell) of constituents.py:78-164 without its zeros: row 0 = the initial condition of the real cells (and whatever
    the caller puts on ghost cells there), every other row = the boundary series of the few ghost cells that have one.  What a run that
    streams its flow field (levels.FlowWindowFeeder) holds instead of the dense array -- (T, ncell) float64 is 86 GB per constituent at
    1 M cells x 10 801 stamps.  Zero stays the "no boundary value" sentinel (transport.py:258-264)."""

    def __init__(self, n_times: int, n_cells: int, initial_row, ghost_cells, series):
        self.shape = (int(n_times), int(n_cells))
        self.initial_row = np.ascontiguousarray(initial_row, dtype=np.float64).reshape(self.shape[1])
        self.ghost_cells = np.ascontiguousarray(ghost_cells, dtype=np.int64).ravel()
        self.series = np.ascontiguousarray(series, dtype=np.float64).reshape(self.shape[0], len(self.ghost_cells))

    def row(self, t: int) -> np.ndarray:
        out = self.initial_row.copy() if t == 0 else np.zeros(self.shape[1])
        out[self.ghost_cells] = self.series[t]
        return out

    def ghost_block(self, t0: int, t1: int, n_real: int) -> np.ndarray:
        """(t1 - t0, ncell - n_real) dense boundary values of levels [t0, t1)."""
        out = np.zeros((t1 - t0, self.shape[1] - n_real))
        if t0 == 0 and t1 > 0:
            out[0] = self.initial_row[n_real:]
        out[:, self.ghost_cells - n_real] = self.series[t0:t1]
        return out

    def dense(self) -> np.ndarray:
        return np.stack([self.row(t) for t in range(self.shape[0])])


def _is_sparse(a) -> bool:
    return hasattr(a, 'ghost_block')


def _ghost_levels(input_array, t0: int, t1: int, n_real: int) -> np.ndarray:
    return input_array.ghost_block(t0, t1, n_real) if _is_sparse(input_array) else input_array[t0:t1, n_real:]


def _real_input_levels(input_array, n_real: int):
    """Levels whose row has non-zero entries on REAL cells (the IC row; point sources at later levels only in a dense array)."""
    if _is_sparse(input_array):
        return [0] if np.any(input_array.initial_row[:n_real] != 0) else []
    return [int(r) for r in np.nonzero(np.any(input_array[:, :n_real] != 0, axis=1))[0]]


def _real_row(input_array, t: int, n_real: int) -> np.ndarray:
    return (input_array.initial_row[:n_real] if t == 0 else np.zeros(n_real)) if _is_sparse(input_array) else input_array[t, :n_real]


class Constituent:
    """constituents.py:17-75 over arrays: NaN-initialised (T, ncell) state, input_array with the
    initial condition in row 0 and boundary values in ghost-cell columns, three (T, E) flux arrays."""

    def __init__(self, name: str, mesh: Mesh, input_array: np.ndarray, units: str = 'Unknown',
                 store_history: bool = True, state_view: Optional[np.ndarray] = None, flux_views=None):
        """state_view / flux_views: rows of the facade's (T, K, ncell) / (T, 3, K, nedge) history blocks (one device
        snapshot per step lands in them without a host copy); allocated here when absent."""
        T = len(mesh['time'])
        E = len(mesh[EDGES_FACE1])
        ncell = len(mesh['face_x'])
        self.name = name
        self.units = units
        self.input_array = input_array if _is_sparse(input_array) else np.ascontiguousarray(input_array, dtype=np.float64)
        if tuple(self.input_array.shape) != (T, ncell):
            raise ValueError(f'input_array of {name}: expected {(T, ncell)}, got {self.input_array.shape}')
        if store_history:
            if flux_views is not None:
                self.advection_mass_flux, self.diffusion_mass_flux, self.total_mass_flux = flux_views
            else:
                self.advection_mass_flux = np.zeros((T, E))
                self.diffusion_mass_flux = np.zeros((T, E))
                self.total_mass_flux = np.zeros((T, E))
            state = state_view if state_view is not None else np.full((T, ncell), np.nan)
        else:
            self.advection_mass_flux = self.diffusion_mass_flux = self.total_mass_flux = None
            state = state_view if state_view is not None else np.full((2, ncell), np.nan)      # rolling pair of levels
        state[0] = self.input_array.row(0) if _is_sparse(self.input_array) else self.input_array[0]               # constituents.py:94-98
        mesh[name] = state
        self.max_value = None
        self.min_value = None


def input_array_from_tables(mesh: Mesh, ic_cell_index, ic_concentration, bc_by_ghost_cell: Optional[dict] = None):
    """constituents.py:78-98 and :153-164 given already time-aligned boundary series:
    ``bc_by_ghost_cell`` maps ghost-cell id -> (T,) concentrations."""
    T = len(mesh['time'])
    ncell = len(mesh['face_x'])
    arr = np.zeros((T, ncell))
    arr[0, np.asarray(ic_cell_index, dtype=np.int64)] = np.asarray(ic_concentration, dtype=np.float64)
    for g, series in (bc_by_ghost_cell or {}).items():
        arr[:, int(g)] = np.asarray(series, dtype=np.float64)
    return arr


def _boundary_face_table(boundary_faces) -> Dict[str, np.ndarray]:
    """Name -> face ids from either form of mesh.attrs['boundary_data']: the reference's DataFrame (columns 'Name',
    'Face Index'; io/hdf.py:355-436) or a plain {name: faces} dict."""
    if boundary_faces is None:
        return {}
    if hasattr(boundary_faces, 'groupby'):                       # pandas DataFrame
        return {str(name): grp['Face Index'].to_numpy(dtype=np.int64) for name, grp in boundary_faces.groupby('Name', sort=False)}
    return {str(k): np.asarray(v, dtype=np.int64).ravel() for k, v in boundary_faces.items()}


def boundary_series(model_time, bc_frame) -> Dict[str, np.ndarray]:
    """constituents.py:121-150 per boundary line, without the reference's growing DataFrame: the CSV rows of one line
    are merged BACKWARD onto the model stamps (pd.merge_asof: the last row at or before each stamp) and gaps are filled
    by linear interpolation in the row index (Series.interpolate(method='linear'): leading NaNs stay NaN, trailing
    ones repeat the last value).  Returns {line name: (T,) concentrations}."""
    import pandas as pd
    t = pd.DatetimeIndex(np.asarray(model_time)).values.astype('datetime64[ns]').astype(np.int64)
    out = {}
    for name, grp in bc_frame.groupby('RAS2D_TS_Name'):
        grp = grp.sort_values('Datetime', kind='stable')
        gt = pd.DatetimeIndex(grp['Datetime']).values.astype('datetime64[ns]').astype(np.int64)
        gv = grp['Concentration'].to_numpy(dtype=np.float64)
        pos = np.searchsorted(gt, t, side='right') - 1           # merge_asof(direction='backward')
        vals = np.where(pos >= 0, gv[np.maximum(pos, 0)], np.nan)
        ok = ~np.isnan(vals)
        if ok.any() and not ok.all():                             # interpolate(method='linear', limit_direction='forward')
            idx = np.arange(len(vals), dtype=np.float64)
            first = int(np.argmax(ok))
            filled = np.interp(idx, idx[ok], vals[ok])
            filled[:first] = np.nan
            vals = filled
        out[str(name)] = vals
    return out


def input_array_from_csv(mesh: Mesh, initial_conditions_csv: str, boundary_conditions_csv: str,
                        boundary_faces, sparse: bool = False):
    """constituents.py:78-164: IC CSV (Cell_Index, Concentration) -> row 0; BC CSV (RAS2D_TS_Name, Datetime,
    Concentration) time-aligned per boundary line (boundary_series) and written at [every time index, ghost cell of
    every face of the line] in one vectorised assignment per line.  ``boundary_faces`` is mesh.attrs['boundary_data']
    (the reference's DataFrame, io/hdf.py:355-436) or a {name: face ids} dict.  The reference builds a (T x faces)-row
    DataFrame by repeated concat + merge for this (12 s on the Ohio River model, examples/Ohio River.ipynb cell[13]);
    here it is O(T) per line plus the assignment.  NaN stays NaN, as in the reference."""
    import pandas as pd
    T = len(mesh['time'])
    ncell = len(mesh['face_x'])
    ic = pd.read_csv(initial_conditions_csv)
    bc = pd.read_csv(boundary_conditions_csv, parse_dates=['Datetime']).dropna(how='all')
    f2 = np.asarray(mesh[EDGES_FACE2])
    table = _boundary_face_table(boundary_faces)
    if sparse:
        # (sparse=True: the same array as a SparseInputArray -- row 0 + the series of the boundary lines' ghost cells; O(T x boundary cells))
        row0 = np.zeros(ncell)
        row0[ic['Cell_Index'].astype(int).to_numpy()] = ic['Concentration'].to_numpy(dtype=np.float64)
        cells, cols = [], []
        for name, vals in boundary_series(mesh['time'], bc).items():
            faces = table.get(name)
            if faces is None or len(faces) == 0:
                continue
            for g in f2[faces]:                                  # (a later line overwrites an earlier one on a shared ghost cell, as the dense assignment does)
                if int(g) in cells:
                    cols[cells.index(int(g))] = vals
                else:
                    cells.append(int(g)); cols.append(vals)
        series = np.stack(cols, axis=1) if cols else np.zeros((T, 0))
        return SparseInputArray(T, ncell, row0, cells, series)
    arr = np.zeros((T, ncell))
    arr[0, ic['Cell_Index'].astype(int).to_numpy()] = ic['Concentration'].to_numpy(dtype=np.float64)
    for name, vals in boundary_series(mesh['time'], bc).items():
        faces = table.get(name)
        if faces is None or len(faces) == 0:
            continue                                             # a CSV line the flow field does not have: the left merge drops it
        arr[:, f2[faces]] = vals[:, None]
    return arr


_LIVE_MODELS = None


def _close_all_models():
    """atexit hook (registered by _track): every facade still alive releases its page locks and rings while the HIP runtime is
    up and -- the point -- while the pages of its history blocks are still mapped ("process exit")."""
    for mdl in list(_LIVE_MODELS or ()):
        try:
            mdl.close_output()
        except Exception:
            pass


def _track(model):
    global _LIVE_MODELS
    if _LIVE_MODELS is None:
        import atexit
        import weakref
        _LIVE_MODELS = weakref.WeakSet()
        atexit.register(_close_all_models)
    _LIVE_MODELS.add(model)


class ClearwaterRiverine:
    """Drop-in for the transport path of the reference class of the same name
    (transport.py:68-276).  Construct either from arrays (``mesh=`` + ``input_arrays=``) or, when h5py
    is importable, from a HEC-RAS HDF path exactly as the reference does."""

    def __init__(self, flow_field_file_path: Optional[str] = None,
                 diffusion_coefficient_input: Optional[float] = None,
                 constituent_dict: Optional[Dict[str, Dict[str, Any]]] = None,
                 config_filepath: Optional[str] = None, verbose: Optional[bool] = False,
                 datetime_range=None, mesh_file_path: Optional[str] = None, *,
                 mesh: Optional[dict] = None, input_arrays: Optional[Dict[str, np.ndarray]] = None,
                 device: int = 0, tol: float = 1e-12, max_iter: int = 5000, store_history: bool = True,
                 solver: str = 'auto', renumber: bool = True, output_store: Optional[str] = None,
                 output_flux: bool = False, host_state: bool = True, deterministic: bool = False,
                 flow_window: Optional[int] = None):
        self.gdf = None
        self.time_step = 0                                       # transport.py:102
        self.verbose = bool(verbose)
        self.tol = float(tol)
        self.max_iter = int(max_iter)
        self.store_history = bool(store_history)
        self.solver = solver
        # deterministic=True: every step runs the ping-pong passes (CWR_STEP_DETERMINISTIC): bitwise reproducible from run to run,
        # like the reference's spsolve; the default chained in-place passes agree with them to <= 1e-10
        self.deterministic = bool(deterministic)
        if mesh is None:
            if mesh_file_path:
                raise NotImplementedError('loading a saved zarr/netCDF mesh is post-processing only in the '
                                          'reference (transport.py:126-139) and outside the transport path')
            if config_filepath or flow_field_file_path:
                from .hdf_reader import read_ras_hdf              # needs h5py; gated
                if config_filepath:
                    import yaml
                    with open(config_filepath) as fh:
                        cfg = yaml.safe_load(fh)
                    if diffusion_coefficient_input is None:
                        diffusion_coefficient_input = cfg['diffusion_coefficient']
                    flow_field_file_path = flow_field_file_path or cfg['flow_field_filepath']
                    constituent_dict = cfg['constituents']
                # flow_window given: the file's levels are read W / 2 at a time while the run goes (hdf_reader lazy=True, levels.FlowWindowFeeder):
                # host memory O(W) levels instead of the whole datetime_range (io/hdf.py:149-191 reads it whole)
                mesh = read_ras_hdf(flow_field_file_path, datetime_range=datetime_range, lazy=flow_window is not None)
            else:
                raise TypeError('Missing a `config_filepath` or a `constituent_dict` and '
                                '`flow_field_file_path` to run the model.')      # transport.py:121-123
        m = mesh if isinstance(mesh, Mesh) else Mesh(mesh)
        if diffusion_coefficient_input is not None:
            m.attrs['diffusion_coefficient'] = float(diffusion_coefficient_input)
        elif 'diffusion_coefficient' in m:
            m.attrs['diffusion_coefficient'] = float(m['diffusion_coefficient'])
        if 'diffusion_coefficient' not in m.attrs:
            raise TypeError('diffusion_coefficient_input is required')
        if 'time' not in m:
            m['time'] = np.asarray(m['time_seconds'], dtype=np.float64)
        f1 = np.ascontiguousarray(m[EDGES_FACE1], dtype=np.int32)
        f2 = np.ascontiguousarray(m[EDGES_FACE2], dtype=np.int32)
        m.attrs[NUMBER_OF_REAL_CELLS] = int(f1.max())            # io/hdf.py:268-269
        m.attrs.setdefault('boundary_data', m.get('boundary_data'))
        self.mesh = m
        self.boundary_data = m.attrs['boundary_data']
        n = m.attrs[NUMBER_OF_REAL_CELLS] + 1
        ncell = len(m['face_x'])
        T = len(m['time'])
        self._n, self._ncell, self._T = n, ncell, T

        # a LEVEL SOURCE instead of the three (T, .) arrays (levels.py: `.read(t0, t1)` or a callable): the flow field is streamed
        level_source = m.attrs.get('level_source')
        if level_source is None and 'level_source' in m:
            level_source = m.pop('level_source')
        if level_source is not None and flow_window is None:
            flow_window = 16
        # a-1 host part: centroid distances and dt; the rest is derived on the GPU
        m[FACE_TO_FACE_DISTANCE] = face_to_face_distance(m)
        m[CHANGE_IN_TIME] = change_in_time(m['time'])

        # constituents (transport.py:158-199)
        if input_arrays is None:
            if not isinstance(constituent_dict, dict):
                raise TypeError('Missing a `config_filepath` or a `constituent_dict` and '
                                '`flow_field_file_path` to run the model.')
            bfaces = m.attrs.get('boundary_data')
            if bfaces is None:
                bfaces = m.attrs.get('boundary_faces') or {}
            input_arrays = {
                name: input_array_from_csv(m, cfg['initial_conditions'], cfg['boundary_conditions'], bfaces, sparse=level_source is not None)
                for name, cfg in constituent_dict.items()}
        self.constituents = list(input_arrays.keys())
        self.constituent_dict: Dict[str, Constituent] = {}
        K = len(self.constituents)
        # History blocks: mesh[name] is the (T, ncell) view [:, k, :] of ONE (T, K, ncell) array and the flux arrays are views of
        # ONE (T, 3, K, nedge) array, so that a step's device snapshot -- constituent-major, exactly (K, ncell) | (3, K, nedge) --
        # is copied by the GPU's DMA engine straight into row t+1 / row t (cwr_output_push_into) with no host copy in between.
        E = len(f1)
        self._state_block = _page_block((T if self.store_history else 2, K, ncell), np.nan)
        self._flux_block = _page_block((T, 3, K, E), 0.0) if self.store_history else None
        self._pinned = []
        for k, name in enumerate(self.constituents):
            units = (constituent_dict or {}).get(name, {}).get('units', 'Unknown') if constituent_dict else 'Unknown'
            fv = None if self._flux_block is None else tuple(self._flux_block[:, q, k, :] for q in range(3))
            self.constituent_dict[name] = Constituent(name, m, input_arrays[name], units, self.store_history,
                                                      state_view=self._state_block[:, k, :], flux_views=fv)

        # engine: topology, flow field and boundary values resident in HBM
        # meshes beyond one workgroup of the one-launch solver (> 4 096 cells): internal space-filling-curve numbering (ordering.py),
        # which is what keeps the 64-row tiles of the sweep kernels compact; reference ids stay at this boundary
        # (up to 24 576 cells the engine takes the one-launch solver with several workgroups and an order of its own, derived from
        # the adjacency; the numbering here then only serves the step's other kernels -- and the tiled passes as the fallback)
        # (within every tile-sized window of the curve the cells are sorted by their work: ordering.balance_windows)
        # (lanes along the principal flow axis, which the engine's chained passes walk: ordering.lane_order; CWR_NO_CHAINS=1 or
        # CWR_TILE_ORDER=hilbert: the isotropic Hilbert curve of round 2)
        curve = None
        if renumber and n > 4096:
            from .distributed import curve_kind                  # (one rule for the facade and the partitioned engines)
            lanes = curve_kind(n, K, 1) == 'lanes'
            if lanes and FLOW_ACROSS_FACE not in m:
                # a streamed field: the lanes follow the flow of its first levels (the numbering is a performance choice only)
                from .levels import as_level_source
                ff0 = as_level_source(level_source, T, len(f1), ncell).read(0, min(T, 8))[0]
                view = Mesh(m); view.attrs = m.attrs; view[FLOW_ACROSS_FACE] = np.asarray(ff0, dtype=np.float32)
                curve = lane_order(view, n, tile_rows=tile_rows(K))
            else:
                curve = lane_order(m, n, tile_rows=tile_rows(K)) if lanes else hilbert_order(m['face_x'], m['face_y'], n)
        order = balance_windows(curve, f1, f2, window=tile_rows(K)) if curve is not None else None
        self.engine = TransportEngine(f1, f2, ncell, K, device=device, cell_order=order)
        # Flow field: all T levels resident in HBM, or -- flow_window=W, or by itself when T levels exceed CWR_FLOW_RESIDENT_LIMIT_MB
        # (default 65 536) of device memory -- a ring of W levels that update() refills one level per step on the engine's flow stream,
        # beside the steps (cwr_flow_window_open / _load: SURVEY 8 f-1; the reference windows a file by datetime_range, io/hdf.py:149-191,
        # and its own fixture has 10 801 stamps).  Same results, bit for bit with deterministic=True.
        import os
        per_level = E * 16 + ncell * 4
        limit = int(os.environ.get('CWR_FLOW_RESIDENT_LIMIT_MB', '65536')) << 20
        if flow_window is None and T * per_level > limit:
            flow_window = max(4, int(limit // per_level))
        self._flow_window = None if flow_window is None else max(2, min(int(flow_window), T))
        self._win_hi = 0                                         # levels [.., _win_hi) have been handed to the engine
        self._feeder = None
        n_g = ncell - n
        if self._flow_window is None:
            self.engine.load_flow_field(m[FLOW_ACROSS_FACE], m[EDGE_VELOCITY], m[VOLUME], m[CHANGE_IN_TIME],
                                        m[FACE_TO_FACE_DISTANCE], m.attrs['diffusion_coefficient'])
        elif level_source is not None:
            # file -> two page-locked staging blocks -> ring (levels.FlowWindowFeeder): W / 2 levels per read, the boundary values of the
            # same levels with them; nothing of length T but the stamps, dt and what the caller's input arrays hold
            from .levels import FlowWindowFeeder, as_level_source
            src = as_level_source(level_source, T, E, ncell)
            self.engine.flow_window_open(T, self._flow_window, m[CHANGE_IN_TIME], m[FACE_TO_FACE_DISTANCE], m.attrs['diffusion_coefficient'])
            self.engine.alloc_boundary(T)
            cons = [self.constituent_dict[c] for c in self.constituents]

            def boundary_levels(t0, t1, cons=cons, n=n):
                return np.stack([_ghost_levels(c.input_array, t0, t1, n) for c in cons], axis=2)

            # (the volume columns of _mass_bal_global, postproc_util.py:92-130, need face_flow on the boundary-line faces at every level: a few
            # columns, kept as the chunks pass by -- (T, faces on lines) float32 instead of (T, nedge))
            bd0 = m.attrs.get('boundary_data') if m.attrs.get('boundary_data') is not None else m.attrs.get('boundary_faces')
            self._line_face_ids = None
            keep_lines = None
            if bd0 is not None and len(bd0):
                ids = np.unique(np.concatenate([np.asarray(f, dtype=np.int64).ravel() for _, f in boundary_lines(bd0)]))
                self._line_face_ids = ids
                self._line_flow = np.full((T, len(ids)), np.nan, dtype=np.float32)

                def keep_lines(a, b, ff, ev, vol, ids=ids):
                    self._line_flow[a:b] = np.asarray(ff)[:, ids]
            self._feeder = FlowWindowFeeder(self.engine, src, T, self._flow_window, cell_cols=self.engine._cols if order is not None else None,
                                            boundary=boundary_levels if n_g > 0 else None, on_levels=keep_lines)
            self._feeder.fill(0)
        else:
            self._flow_arrays = (np.ascontiguousarray(m[FLOW_ACROSS_FACE], dtype=np.float32), np.ascontiguousarray(m[EDGE_VELOCITY], dtype=np.float32),
                                 self.engine.volume_in_engine_order(m[VOLUME]))          # (volumes permuted to the engine's cell order once)
            self._flow_pinned = [a for a in self._flow_arrays if a.nbytes >= (1 << 20) and self.engine.host_register(a)]   # (asynchronous uploads)
            self.engine.flow_window_open(T, self._flow_window, m[CHANGE_IN_TIME], m[FACE_TO_FACE_DISTANCE], m.attrs['diffusion_coefficient'])
            self._fill_flow_window(0)
        if self._feeder is None:
            ghost = np.stack([_ghost_levels(self.constituent_dict[c].input_array, 0, T, n) for c in self.constituents], axis=2)
            self.engine.load_boundary(ghost)
        # rows of input_array that carry non-zero values in REAL cells (the IC row, normally only t = 0):
        # RHS.update_values overwrites the solution with them (linalg.py:199-200)
        self._real_input_rows = set()
        for c in self.constituents:
            self._real_input_rows.update(_real_input_levels(self.constituent_dict[c].input_array, n))
        # levels >= 1: the reference also writes them into the SOLVED level t+1 before the mass fluxes are taken
        # (transport.py:258-264); the engine does that on the device from these sparse entries
        lv, ce, va = [], [], []
        for r in sorted(self._real_input_rows - {0}):
            inp = np.stack([_real_row(self.constituent_dict[c].input_array, r, n) for c in self.constituents], axis=1)
            cells = np.nonzero(np.any(inp != 0, axis=1))[0]
            lv.append(np.full(len(cells), r, dtype=np.int32)); ce.append(cells); va.append(inp[cells])
        if lv:
            self.engine.load_real_inputs(np.concatenate(lv), np.concatenate(ce), np.concatenate(va))
        self._device_level = -1
        self.last_step: Optional[StepResult] = None

        # output side: device mass-balance ledger over the boundary-condition lines, streamed zarr output
        self.host_state = bool(host_state)
        self._lines = []
        bd = self.boundary_data if self.boundary_data is not None else m.attrs.get('boundary_faces')
        if bd is not None and len(bd):
            self._lines = boundary_lines(bd)
            self.engine.set_boundary_lines([f for _, f in self._lines])
        self._mass_start = None
        self._stream = None
        self._ring = None                                        # pinned one-slot ring of update()'s own read-out (opened lazily)
        _track(self)
        if output_store is not None:
            self._stream = StreamedOutput(self.engine, output_store, self.constituents, T, with_flux=output_flux,
                                          attrs={'diffusion_coefficient': m.attrs['diffusion_coefficient']})

    # ------------------------------------------------------------------ helpers
    def _fill_flow_window(self, t: int):
        """Windowed flow field: hand the engine every level up to t + W - 1 that it does not hold yet (the slot of level L held level
        L - W, which no step from t on reads).  Enqueued on the engine's flow stream: returns at once."""
        if self._flow_window is None:
            return
        if self._feeder is not None:
            self._feeder.fill(t)
            return
        hi = min(self._T, t + self._flow_window)
        if hi > self._win_hi:
            ff, ev, vol = self._flow_arrays
            lo = self._win_hi
            self.engine.flow_window_load(lo, ff[lo:hi], ev[lo:hi], vol[lo:hi], engine_order=True)
            self._win_hi = hi

    def _row(self, name: str, t: int) -> np.ndarray:
        st = self.mesh[name]
        return st[t] if self.store_history else st[t % 2]

    def coefficients(self, t: int):
        """advection_coeff[t] (f32) and coeff_to_diffusion[t] (f64) as derived on the device."""
        return self.engine.get_coefficients(t)

    # ------------------------------------------------------------------ the hot path
    def update(self, update_concentration: Optional[dict] = None, reaction_matrix=None):
        """Update a single timestep (transport.py:201-276).

        reaction_matrix (K, K), optional: apply c[cell, :] <- M c[cell, :] to the level-t state ON THE DEVICE before
        the transport step -- the same effect as passing update_concentration = {name: (M c)[:, k]} but without
        moving the state through the host."""
        t = self.time_step
        n = self._n
        if t + 1 >= self._T:
            raise IndexError(f'time step {t} is the last level of the flow field')
        overridden = False
        if isinstance(update_concentration, dict):
            for cname in update_concentration:
                if cname not in self.constituent_dict:           # transport.py:223-229
                    print(f'WARNING: {cname} is not being used in the model.')
                    print('Please review the constituent names in the update dictionary')
            for cname in self.constituents:
                if cname in update_concentration:                # transport.py:233-236
                    vals = update_concentration[cname]
                    vals = np.asarray(getattr(vals, 'values', vals), dtype=np.float64)
                    self._row(cname, t)[0:n] = vals[0:n]
                    overridden = True
        # (real-cell inputs of levels >= 1 are already in the device state: the engine applied them when it solved level t)
        if overridden or self._device_level != t or (t == 0 and t in self._real_input_rows):
            x = np.stack([self._row(c, t)[0:n] for c in self.constituents], axis=1)
            if t in self._real_input_rows:                       # linalg.py:199-200
                inp = np.stack([_real_row(self.constituent_dict[c].input_array, t, n) for c in self.constituents], axis=1)
                x = np.where(inp != 0, inp, x)
            self.engine.set_state(x)
        if reaction_matrix is not None:
            self.engine.react_linear(reaction_matrix)
            if t in self._real_input_rows:
                # as with the host override, non-zero input_array[t] entries on real cells win over the reaction's result
                # (linalg.py:199-200) -- at level 0 that is the whole initial condition
                x = self.engine.get_state()[:n]
                inp = np.stack([_real_row(self.constituent_dict[c].input_array, t, n) for c in self.constituents], axis=1)
                self.engine.set_state(np.where(inp != 0, inp, x))
            if self.store_history:                               # keep mesh[name][t] consistent, as the override does
                c_now = self.engine.get_state()
                for k, cname in enumerate(self.constituents):
                    self._row(cname, t)[0:n] = c_now[:n, k]
        if self._mass_start is None:                             # postproc_util.py:36-45: mass in the domain at level 0
            self._mass_start = self.engine.domain_mass(t)
            if self._stream is not None:                         # level 0 is host data (input_array row 0)
                for cname in self.constituents:
                    self._stream.writer.write_level(cname, t, np.ascontiguousarray(self._row(cname, t), dtype=np.float64))
        self._fill_flow_window(t)
        want_flux = self.store_history or (self._stream is not None and self._stream.with_flux)
        # (a step that raises leaves the device state at level t -- the engine restores it -- and time_step unchanged:
        # update() may simply be called again, e.g. with a larger max_iter)
        self.last_step = self.engine.step(t, tol=self.tol, max_iter=self.max_iter, mass_flux=want_flux,
                                          solver=self.solver, mass_balance=bool(self._lines), deterministic=self.deterministic)
        self._device_level = t + 1
        if self._stream is not None:
            self._stream.push(t + 1)                             # asynchronous: pinned ring + writer thread
        if not self.host_state:                                  # results live in the store / on the device only
            self.time_step += 1
            return
        if self._ring is None and self._stream is None:
            # One snapshot per step instead of a blocking get_state + three get_mass_flux read-outs and per-constituent strided
            # copies: the engine transposes state (and fluxes) constituent-major on the device (k_snapshot_t) and its copy
            # engine writes them into row t+1 / t of the history blocks.  The blocks are page-locked when the runtime allows
            # (CWR_PIN_LIMIT_MB, default 4096), which makes those copies asynchronous DMA.  At the reference's own sizes
            # (2 943 cells x 12) the facade used to cost 1.02 ms per step around a 0.38 ms engine step.
            import os
            self.engine.output_open(n_slots=1, with_flux=self.store_history)
            self._ring = True
            blocks = [b for b in (self._state_block, self._flux_block) if b is not None]
            if sum(b.nbytes for b in blocks) <= (int(os.environ.get('CWR_PIN_LIMIT_MB', '4096')) << 20):
                self._pinned = [b for b in blocks if self.engine.host_register(b)]
        if self._ring:
            r = (t + 1) if self.store_history else (t + 1) % 2
            slot = self.engine.output_push_into(self._state_block[r], None if self._flux_block is None else self._flux_block[t])
            self.engine.output_wait(slot)                        # transport.py:252-264 (state), :267-273 (fluxes): complete on return
            self.engine.output_release(slot)
            self.time_step += 1                                  # transport.py:276
            return
        c_all = self.engine.get_state()                          # (ncell, K): transport.py:252-264
        for k, cname in enumerate(self.constituents):
            self._row(cname, t + 1)[:] = c_all[:, k]
        if self.store_history:                                   # transport.py:267-273
            adv, dif, tot = self.engine.get_mass_flux()
            for k, cname in enumerate(self.constituents):
                con = self.constituent_dict[cname]
                con.advection_mass_flux[t] = adv[:, k]
                con.diffusion_mass_flux[t] = dif[:, k]
                con.total_mass_flux[t] = tot[:, k]
        self.time_step += 1                                      # transport.py:276

    def simulate_wq(self, *a, **kw):
        """Deprecated and unrunnable in the reference at HEAD (transport.py:279-372 reads ``x`` before
        assignment); kept as the loop the docstring promises: update() over every remaining level."""
        import warnings
        warnings.warn('Use `update` method instead.', DeprecationWarning)
        while self.time_step + 1 < self._T:
            self.update()

    def set_value_range(self, constituent_name: Optional[str] = None):
        """constituents.py:167-172."""
        names = [constituent_name] if constituent_name is not None else self.constituents
        for nme in names:
            real = self.mesh[nme][:, : self._n]
            self.constituent_dict[nme].max_value = int(np.nanmax(real))
            self.constituent_dict[nme].min_value = int(np.nanmin(real))

    def mass_bal_global(self, constituent_name: str) -> dict:
        """postproc_util._mass_bal_global (:21-166) for the levels simulated so far, from the device ledger: a dict
        keyed by the reference's DataFrame columns."""
        if not self._lines:
            raise ValueError('boundary_data_path input required')           # postproc_util.py:367-368
        if self._mass_start is None:
            raise IndexError('no step taken yet')
        k = self.constituents.index(constituent_name)
        m = self.mesh
        if FLOW_ACROSS_FACE not in m:
            # a streamed run: the boundary-line columns the feeder kept (levels not read yet are NaN and skipped, as xarray's sum skips the
            # trailing NaN of dt: postproc_util.py:92-98 sums over the whole window -- call this when the run is over)
            pos = {int(f): i for i, f in enumerate(self._line_face_ids)}
            lines_local = [(nm, np.array([pos[int(f)] for f in faces], dtype=np.int64)) for nm, faces in self._lines]
            vols = volume_columns(self._line_flow, m[CHANGE_IN_TIME], lines_local)
        else:
            vols = volume_columns(m[FLOW_ACROSS_FACE], m[CHANGE_IN_TIME], self._lines)
        mass0, vol0 = self._mass_start
        mass1, vol1 = self.engine.domain_mass(self.time_step)
        return assemble(self._lines, vols, self.engine.get_mass_balance()[:, :, k], vol0, float(mass0[k]), vol1, float(mass1[k]))

    def close_output(self):
        """Drain the streamed-output ring and finish the store; release the page locks of the history blocks."""
        if getattr(self, '_stream', None) is not None:
            self._stream.close()
            self._stream = None
        if getattr(self, '_ring', None):
            if self.engine._h:                                   # (an engine that was closed first took its ring with it)
                self.engine.synchronize()
                self.engine.output_close()
            self._ring = None
        for b in getattr(self, '_pinned', []):
            self.engine.host_unregister(b)
        self._pinned = []
        if getattr(self, '_feeder', None) is not None:
            if self.engine._h:
                self.engine.synchronize()
            self._feeder.close()
            self._feeder = None
        if getattr(self, '_flow_pinned', None):
            if self.engine._h:
                self.engine.synchronize()
            for b in self._flow_pinned:
                self.engine.host_unregister(b)
            self._flow_pinned = []

    def __del__(self):
        # the page locks must go BEFORE the blocks' pages are unmapped: a stale registration of a recycled address range makes
        # later copies fail (or land in the wrong pages).  At interpreter shutdown the atexit hook below has already done it
        # (and the HIP runtime may be gone): nothing to do then.
        try:
            if sys is None or sys.is_finalizing():
                return
            self.close_output()
        except Exception:
            pass

    def finalize(self, save: bool = False, output_filepath: Optional[str] = None):
        """transport.py:385-395: value ranges, then (save=True) the mesh as a zarr store (io/outputs.py:11-17;
        '.zarr' paths) or an .npz archive; netCDF needs a package this image does not have."""
        self.close_output()
        if self.host_state and self.store_history:
            self.set_value_range()
        if save and str(output_filepath).endswith('.zarr'):
            if not self.store_history:
                raise ValueError('no history in RAM: construct with output_store=... to stream the run to zarr')
            arrays = {c: (self._ncell, 'nface') for c in self.constituents}
            w = ZarrStreamWriter(output_filepath, arrays, self._T, {'diffusion_coefficient': self.mesh.attrs['diffusion_coefficient']})
            for c in self.constituents:
                for t in range(self._T):
                    w.write_level(c, t, np.ascontiguousarray(self.mesh[c][t], dtype=np.float64))
        elif save and str(output_filepath).endswith('.nc'):
            raise ValueError('Cannot save as .nc: netCDF output needs the netCDF4 package')
        elif save:
            np.savez_compressed(output_filepath, **{k: np.asarray(v) for k, v in self.mesh.items()
                                                    if isinstance(v, np.ndarray)})
