"""CPU ORACLE -- test infrastructure, NOT product code.

A numpy/scipy restatement of the per-step transport path of
EcohydrologyTeam/ClearWater-riverine (reference @ 2025-07-04, __version__ 0.6.0):
``ClearwaterRiverine.update()`` = LHS COO assembly -> csr_matrix -> per-constituent
RHS assembly -> scipy.sparse.linalg.spsolve -> write-back -> per-edge mass flux.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / reported baseline.  The
product path (``clearwater-riverine_amd/``) never imports it.

Every function cites the reference lines it follows; paths are relative to
``/root/reference/src/clearwater_riverine/``.  The reference indexes an
``xarray.Dataset``; here the same variables (names from ``variables.py:1-37``)
live in a plain dict of numpy arrays (``mesh``):

    mesh['edges_face1'], mesh['edges_face2']   (E,)  int32   io/hdf.py:257-266
    mesh['nreal']                              int           io/hdf.py:268-269  (max(face1); real cells are 0..nreal)
    mesh['face_x'], mesh['face_y']             (ncell,) f64
    mesh['face_flow'], mesh['edge_velocity']   (T,E) f32     io/hdf.py:275-310
    mesh['volume']                             (T,ncell) f32
    mesh['time_seconds']                       (T,) f64      seconds since the first stamp
    mesh['diffusion_coefficient']              python float  mesh.py:19-55 (attrs)
  derived by derive_coefficients() (utilities.py:513-541):
    mesh['advection_coeff'] (T,E) f32, mesh['edge_vertical_area'] (T,E) f32,
    mesh['face_to_face_dist'] (E,) f64, mesh['coeff_to_diffusion'] (T,E) f64, mesh['dt'] (T,) f64

PARITY PINNING.  The reference package cannot be imported in the build image
(ordinary ModuleNotFoundError: xarray / holoviews / geoviews / geopandas are not
installed and there is no network), and its own tests are stale and assert no
numerical output of update().  The oracle is therefore pinned by (tests/test_oracle.py):
  * the reference's fixture facts and notebook-printed known answers
    (time-stamp counts, BC ghost cells 4 and 6 on plan02,
    Mass_start/Mass_end 5000.553131 / 5001.221848, plan03 diffusion sums 0.00300032 ...),
  * the identity "entry-by-entry COO assembly == per-cell operator form" on the
    reference's HDF fixtures, and the discrete mass balance built from _mass_flux.
No reference *output* of update() exists to compare with: on that point parity
is "pinned by restatement", as SURVEY.md section 8c records.

dtype notes (SURVEY.md section 8 "dtype note"): V is float32, dt is a float64 scalar;
V/dt and V*x/dt are evaluated in float64 (NumPy >= 2 / NEP 50 behaviour).
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.linalg import spsolve

# variable names, verbatim from variables.py:12-37
EDGES_FACE1 = 'edges_face1'
EDGES_FACE2 = 'edges_face2'
NUMBER_OF_REAL_CELLS = 'nreal'
VOLUME = 'volume'
EDGE_VELOCITY = 'edge_velocity'
CHANGE_IN_TIME = 'dt'
FLOW_ACROSS_FACE = 'face_flow'
ADVECTION_COEFFICIENT = 'advection_coeff'
EDGE_VERTICAL_AREA = 'edge_vertical_area'
FACE_TO_FACE_DISTANCE = 'face_to_face_dist'
COEFFICIENT_TO_DIFFUSION_TERM = 'coeff_to_diffusion'


# --------------------------------------------------------------------------- a-1
def derive_coefficients(mesh: dict) -> dict:
    """utilities.py:513-541 (else-branch of WQVariableCalculator.calculate),
    _calc_distances_cell_centroids :261-278, _calc_coeff_to_diffusion_term :280-305.

    Adds advection_coeff, edge_vertical_area, face_to_face_dist, coeff_to_diffusion, dt.
    """
    flow = np.asarray(mesh[FLOW_ACROSS_FACE], dtype=np.float32)
    vel = np.asarray(mesh[EDGE_VELOCITY], dtype=np.float32)
    # :514-517  advection_coeff = face_flow * sign(|edge_velocity|)           (float32)
    adv = flow * np.sign(np.abs(vel))
    # :519-523  edge_vertical_area = (advection_coeff / edge_velocity).fillna(0)  (float32)
    with np.errstate(divide='ignore', invalid='ignore'):
        area = adv / vel
    area = np.where(np.isnan(area), np.float32(0.0), area).astype(np.float32)
    # :261-278  distance between the two cell centres of every edge        (float64)
    f1 = np.asarray(mesh[EDGES_FACE1])
    f2 = np.asarray(mesh[EDGES_FACE2])
    fx = np.asarray(mesh['face_x'], dtype=np.float64)
    fy = np.asarray(mesh['face_y'], dtype=np.float64)
    dist = np.sqrt((fx[f1] - fx[f2]) ** 2 + (fy[f1] - fy[f2]) ** 2)
    # :304-305  area * D / dist for ALL edges (the ghost mask built at :294-301 is unused).
    # float32 array * python float stays float32; dividing by the float64 distance promotes.
    D = mesh['diffusion_coefficient']
    with np.errstate(divide='ignore', invalid='ignore'):
        dif = (area * D) / dist[None, :]
    # :537-541  dt = diff(time) in seconds, trailing NaN
    ts = np.asarray(mesh['time_seconds'], dtype=np.float64)
    dt = np.append(np.ediff1d(ts), np.nan)
    mesh[ADVECTION_COEFFICIENT] = adv
    mesh[EDGE_VERTICAL_AREA] = area
    mesh[FACE_TO_FACE_DISTANCE] = dist
    mesh[COEFFICIENT_TO_DIFFUSION_TERM] = np.asarray(dif, dtype=np.float64)
    mesh[CHANGE_IN_TIME] = dt
    return mesh


# --------------------------------------------------------------------------- a-2
class LHS:
    """linalg.py:17-156, entry for entry (duplicate-bearing COO, float row/col arrays)."""

    def __init__(self, mesh: dict):
        f1 = np.asarray(mesh[EDGES_FACE1])
        f2 = np.asarray(mesh[EDGES_FACE2])
        nreal = mesh[NUMBER_OF_REAL_CELLS]
        self.internal_edges = np.where((f1 <= nreal) & (f2 <= nreal))[0]      # :28
        self.internal_edge_count = len(self.internal_edges)                   # :29
        self.real_edges_face1 = np.where(f1 <= nreal)[0]                      # :30
        self.real_edges_face2 = np.where(f2 <= nreal)[0]                      # :31
        self.nreal_count = nreal + 1                                          # :32

    def update_values(self, mesh: dict, t: int):
        f1 = np.asarray(mesh[EDGES_FACE1])
        f2 = np.asarray(mesh[EDGES_FACE2])
        a_t = mesh[ADVECTION_COEFFICIENT][t]
        d_t = mesh[COEFFICIENT_TO_DIFFUSION_TERM][t]
        n = self.nreal_count
        nedge = np.arange(len(f1))
        is_internal = np.isin(nedge, self.internal_edges)
        flow_out_indices = np.where(a_t > 0)[0]                               # :61
        flow_out_indices_internal = np.where((a_t > 0) & is_internal)[0]      # :62-63
        flow_in_indices = np.where((a_t < 0) & is_internal)[0]                # :64-65
        v_next = mesh[VOLUME][t + 1]
        empty_cells = np.where((v_next == 0) & (np.arange(len(v_next)) < n))[0][0:n]   # :66

        len_val = self.internal_edge_count * 2 + n * 2 + \
            len(flow_out_indices) * 2 + len(flow_in_indices) * 2 + len(empty_cells) + \
            len(self.real_edges_face1) + len(self.real_edges_face2)          # :69-71
        self.rows = np.zeros(len_val)
        self.cols = np.zeros(len_val)
        self.coef = np.zeros(len_val)

        # :76-81 dummy 1 on the diagonal of dry cells
        start = 0
        end = len(empty_cells)
        self.rows[start:end] = empty_cells
        self.cols[start:end] = empty_cells
        self.coef[start:end] = 1

        # :84-89 V[t+1]/dt on the diagonal (float64 division, see module docstring)
        start = end
        end = end + n
        self.rows[start:end] = np.arange(n)
        self.cols[start:end] = np.arange(n)
        seconds = mesh[CHANGE_IN_TIME][t]
        self.coef[start:end] = v_next[0:n].astype(np.float64) / seconds

        # :92-103 diffusion on the diagonal of face1 (if real) and face2 (if real)
        start = end
        end = end + len(self.real_edges_face1)
        self.rows[start:end] = f1[self.real_edges_face1]
        self.cols[start:end] = f1[self.real_edges_face1]
        self.coef[start:end] = d_t[self.real_edges_face1]
        start = end
        end = end + len(self.real_edges_face2)
        self.rows[start:end] = f2[self.real_edges_face2]
        self.cols[start:end] = f2[self.real_edges_face2]
        self.coef[start:end] = d_t[self.real_edges_face2]

        # :107-122 outflow (a > 0): +a on diag of P for EVERY such edge, -a at (N, P) for internal ones
        if len(flow_out_indices) > 0:
            start = end
            end = end + len(flow_out_indices)
            self.rows[start:end] = f1[flow_out_indices]
            self.cols[start:end] = f1[flow_out_indices]
            self.coef[start:end] = a_t[flow_out_indices]
            start = end
            end = end + len(flow_out_indices_internal)
            self.rows[start:end] = f2[flow_out_indices_internal]
            self.cols[start:end] = f1[flow_out_indices_internal]
            self.coef[start:end] = a_t[flow_out_indices_internal] * -1

        # :124-141 inflow (a < 0, internal only): +a at (P, N), -a on diag of N
        if len(flow_in_indices) > 0:
            start = end
            end = end + len(flow_in_indices)
            self.rows[start:end] = f1[flow_in_indices]
            self.cols[start:end] = f2[flow_in_indices]
            self.coef[start:end] = a_t[flow_in_indices]
            start = end
            end = end + len(flow_in_indices)
            self.rows[start:end] = f2[flow_in_indices]
            self.cols[start:end] = f2[flow_in_indices]
            self.coef[start:end] = a_t[flow_in_indices] * -1

        # :145-156 -d off the diagonal, both ways, internal edges
        start = end
        end = end + self.internal_edge_count
        self.rows[start:end] = f1[self.internal_edges]
        self.cols[start:end] = f2[self.internal_edges]
        self.coef[start:end] = -1 * d_t[self.internal_edges]
        start = end
        end = end + self.internal_edge_count
        self.rows[start:end] = f2[self.internal_edges]
        self.cols[start:end] = f1[self.internal_edges]
        self.coef[start:end] = -1 * d_t[self.internal_edges]

    def csr(self) -> csr_matrix:
        """transport.py:215-218."""
        n = self.nreal_count
        return csr_matrix((self.coef, (self.rows, self.cols)), shape=(n, n))


def apply_percell(mesh: dict, t: int, x: np.ndarray) -> np.ndarray:
    """The algebraically equivalent per-cell ("face-flux") form of A.x that the HIP
    kernel implements.  x is (n,) or (n, K).  Loop-free numpy,
    accumulating with np.add.at in edge order.
    """
    f1 = np.asarray(mesh[EDGES_FACE1]).astype(np.int64)
    f2 = np.asarray(mesh[EDGES_FACE2]).astype(np.int64)
    nreal = mesh[NUMBER_OF_REAL_CELLS]
    n = nreal + 1
    x2 = x.reshape(n, -1).astype(np.float64)
    a = mesh[ADVECTION_COEFFICIENT][t].astype(np.float64)
    d = mesh[COEFFICIENT_TO_DIFFUSION_TERM][t].astype(np.float64)
    v_next = mesh[VOLUME][t + 1][0:n].astype(np.float64)
    dt = mesh[CHANGE_IN_TIME][t]
    diag = v_next / dt + (v_next == 0)
    y = diag[:, None] * x2
    p_real = f1 <= nreal
    n_real = f2 <= nreal
    internal = p_real & n_real
    ap = np.maximum(a, 0.0)
    am = np.minimum(a, 0.0)
    # rows of P (P real): d*(x[P] - x[N][N real]) + max(a,0)*x[P] + min(a,0)*x[N][internal]
    e = np.where(p_real)[0]
    xn = np.where(n_real[e, None], x2[np.minimum(f2[e], nreal)], 0.0)
    contrib = d[e, None] * (x2[f1[e]] - xn) + ap[e, None] * x2[f1[e]] + \
        np.where(internal[e, None], am[e, None] * xn, 0.0)
    np.add.at(y, f1[e], contrib)
    # rows of N (N real): d*x[N] always; internal: -d*x[P] - max(a,0)*x[P] - min(a,0)*x[N]
    e = np.where(n_real)[0]
    xp = np.where(p_real[e, None], x2[np.minimum(f1[e], nreal)], 0.0)
    contrib = d[e, None] * x2[f2[e]] + np.where(
        internal[e, None],
        -d[e, None] * xp - ap[e, None] * xp - am[e, None] * x2[f2[e]], 0.0)
    np.add.at(y, f2[e], contrib)
    return y.reshape(x.shape)


# --------------------------------------------------------------------------- a-3
class RHS:
    """linalg.py:158-406."""

    def __init__(self, mesh: dict, input_array: np.ndarray):
        self.nreal_count = mesh[NUMBER_OF_REAL_CELLS] + 1                    # :172
        self.input_array = input_array                                       # :173
        self.vals = np.zeros(self.nreal_count)                               # :174
        self.ghost_cells = np.where(np.asarray(mesh[EDGES_FACE2]) > mesh[NUMBER_OF_REAL_CELLS])[0]  # :175

    def update_values(self, solution: np.ndarray, mesh: dict, t: int):
        ncell = len(mesh['face_x'])
        solver = np.zeros(ncell)                                             # :194-198
        solver[0:self.nreal_count] = solution                                # :199
        nz = self.input_array[t].nonzero()
        solver[nz] = self.input_array[t][nz]                                 # :200
        self.vals[:] = self._calculate_rhs(mesh, t, solver[0:self.nreal_count])  # :201

    def _calculate_load(self, mesh, t, concentrations):
        volume = mesh[VOLUME][t][0:self.nreal_count]                         # :225
        delta_time = mesh[CHANGE_IN_TIME][t]                                 # :213
        return volume.astype(np.float64) * concentrations / delta_time      # :240

    def _calculate_rhs(self, mesh, t, concentrations):
        load = self._calculate_load(mesh, t, concentrations)                 # :273
        n = self.nreal_count
        ghost_cells_in = self._ghost_cell(mesh, t + 1, flowing_in=True)[0:n]     # :274, :258
        ghost_cells_out = self._ghost_cell(mesh, t + 1, flowing_in=False)[0:n]   # :259
        return load + ghost_cells_in + ghost_cells_out                       # :275

    @staticmethod
    def _edge_to_face(edge_array, face_array, mesh_array, index_list, internal_cell_index):
        """:336-352 -- assignment (last write wins), with the `!= 0` filter that makes an
        active ghost edge with a zero coefficient a shape-mismatch ValueError."""
        edge_array[index_list] = np.abs(mesh_array[index_list])              # :349
        values = np.where(edge_array != 0)[0]                                # :350
        face_array[np.array(internal_cell_index)] = edge_array[values]       # :351
        return face_array

    def _ghost_cell(self, mesh, t, flowing_in: bool):
        f1 = np.asarray(mesh[EDGES_FACE1])
        f2 = np.asarray(mesh[EDGES_FACE2])
        nedge = len(f1)
        ncell = len(mesh['face_x'])
        # :278-309
        advection = bool(flowing_in)
        condition = np.less if flowing_in else np.greater
        advection_edge = np.zeros(nedge) if advection else None              # :329-333
        advection_face = np.zeros(ncell) if advection else None
        diffusion_edge = np.zeros(nedge)
        diffusion_face = np.zeros(ncell)

        velocity_indices = np.where(condition(mesh[EDGE_VELOCITY][t], 0))[0]     # :372
        index_list = np.intersect1d(velocity_indices, self.ghost_cells)      # :373 (sorted)
        internal_cell_index = f1[index_list]                                 # :374
        external_cell_index = f2[index_list]                                 # :375
        concentration_multipliers = np.zeros(ncell)                          # :377
        concentration_multipliers[internal_cell_index] = self.input_array[t][external_cell_index]  # :378

        if len(index_list) != 0:                                             # :380
            if advection:
                advection_face[:] = self._edge_to_face(
                    advection_edge, advection_face, mesh[ADVECTION_COEFFICIENT][t],
                    index_list, internal_cell_index)
            if mesh['diffusion_coefficient'] != 0:                           # :390
                diffusion_face[:] = self._edge_to_face(
                    diffusion_edge, diffusion_face, mesh[COEFFICIENT_TO_DIFFUSION_TERM][t],
                    index_list, internal_cell_index)
        if flowing_in:
            add_to_rhs = advection_face + diffusion_face                     # :400
        else:
            add_to_rhs = diffusion_face                                      # :402
        return add_to_rhs * concentration_multipliers                        # :404


def rhs_percell(mesh: dict, t: int, x_t: np.ndarray, ghost_conc_next: np.ndarray) -> np.ndarray:
    """Per-cell form of b that the HIP kernel implements:
    b[c] = V[t,c]*x[c]/dt[t] + G_in[c] + G_out[c], boundary terms at level t+1, the
    highest active ghost-edge id winning per cell and per set.  x_t is (n,) or (n,K);
    ghost_conc_next is input_array[t+1] as (ncell,) or (ncell,K).  Valid under the
    precondition that active ghost edges have non-zero coefficients.
    """
    f1 = np.asarray(mesh[EDGES_FACE1]).astype(np.int64)
    f2 = np.asarray(mesh[EDGES_FACE2]).astype(np.int64)
    nreal = mesh[NUMBER_OF_REAL_CELLS]
    n = nreal + 1
    x2 = x_t.reshape(n, -1).astype(np.float64)
    g2 = ghost_conc_next.reshape(len(mesh['face_x']), -1)
    b = mesh[VOLUME][t][0:n].astype(np.float64)[:, None] * x2 / mesh[CHANGE_IN_TIME][t]
    vel = mesh[EDGE_VELOCITY][t + 1]
    a = np.abs(mesh[ADVECTION_COEFFICIENT][t + 1].astype(np.float64))
    d = np.abs(mesh[COEFFICIENT_TO_DIFFUSION_TERM][t + 1].astype(np.float64))
    if mesh['diffusion_coefficient'] == 0:
        d = np.zeros_like(d)
    gin = np.zeros_like(b)
    gout = np.zeros_like(b)
    for e in np.where(f2 > nreal)[0]:                 # ascending edge id: last write wins
        if vel[e] < 0:
            gin[f1[e]] = (a[e] + d[e]) * g2[f2[e]]
        elif vel[e] > 0:
            gout[f1[e]] = d[e] * g2[f2[e]]
    return (b + gin + gout).reshape(x_t.shape)


# --------------------------------------------------------------------------- a-6
def mass_flux(mesh: dict, c_next: np.ndarray, t: int):
    """transport.py:406-429.  c_next is the full (ncell,) state at level t+1 (NaN in
    ghost cells without a boundary value).  Returns (advection, diffusion, total) (E,)."""
    f1 = np.asarray(mesh[EDGES_FACE1])
    f2 = np.asarray(mesh[EDGES_FACE2])
    a = mesh[ADVECTION_COEFFICIENT][t]
    negative_condition = a < 0                                               # :414
    parent = c_next[f1]                                                      # :415
    neighbor = c_next[f2]                                                    # :416
    delta_time = mesh[CHANGE_IN_TIME][t]                                     # :417
    adv = np.where(negative_condition, a * neighbor, a * parent) * delta_time      # :419-423
    dif = mesh[COEFFICIENT_TO_DIFFUSION_TERM][t] * (neighbor - parent) * delta_time  # :425-427
    return adv, dif, adv + dif                                               # :429


# --------------------------------------------------------------------------- a-5 / a-7 / a-8
class Constituent:
    """constituents.py:17-75 restated over arrays: the (T, ncell) NaN-initialised state,
    input_array with IC in row 0 and BC values in ghost-cell columns, its RHS and the three
    (T, E) mass-flux arrays."""

    def __init__(self, name: str, mesh: dict, input_array: np.ndarray):
        T = len(mesh['time_seconds'])
        E = len(mesh[EDGES_FACE1])
        ncell = len(mesh['face_x'])
        self.name = name
        self.advection_mass_flux = np.zeros((T, E))                          # :28-30
        self.diffusion_mass_flux = np.zeros((T, E))
        self.total_mass_flux = np.zeros((T, E))
        self.input_array = np.asarray(input_array, dtype=np.float64)         # :31
        assert self.input_array.shape == (T, ncell)
        self.state = np.full((T, ncell), np.nan)                             # :39-48
        self.state[0] = self.input_array[0]                                  # :94-98
        self.b = RHS(mesh, self.input_array)                                 # :62-65


def build_input_array(mesh: dict, ic_cell_index, ic_concentration, bc_ghost_values: dict | None = None):
    """constituents.py:78-98 (IC -> row 0) and :153-164 (BC value at [time index, ghost cell]).
    bc_ghost_values maps ghost-cell id -> (T,) array of concentrations (already time-aligned;
    the CSV merge_asof/interpolate plumbing of :121-150 is outside the hot path)."""
    T = len(mesh['time_seconds'])
    ncell = len(mesh['face_x'])
    arr = np.zeros((T, ncell))
    arr[0, np.asarray(ic_cell_index, dtype=np.int64)] = np.asarray(ic_concentration, dtype=np.float64)
    if bc_ghost_values:
        for g, series in bc_ghost_values.items():
            arr[:, int(g)] = np.asarray(series, dtype=np.float64)
    return arr


def set_boundary_conditions_literal(input_array: np.ndarray, mesh: dict, bc_df, flow_field_boundaries) -> np.ndarray:
    """constituents.py:100-164 restated statement for statement with pandas (the reference's own tool here): per
    RAS2D_TS_Name group merge_asof onto the model stamps + linear interpolation, the groups concatenated, a LEFT merge
    with the flow field's boundary DataFrame on the line name, then the fancy assignment
    input_array[[Time Index], [Ghost Cell]] = Concentration.  bc_df: DataFrame (RAS2D_TS_Name, Datetime, Concentration);
    flow_field_boundaries: DataFrame with 'Name' and 'Face Index' (mesh.attrs['boundary_data'], io/hdf.py:355-436);
    mesh['time'] datetime64 stamps.  Test infrastructure: the product's vectorised pipeline is checked against this."""
    import pandas as pd
    xarray_time_index = pd.DatetimeIndex(np.asarray(mesh['time']))                    # :125-127
    model_dataframe = pd.DataFrame({'Datetime': xarray_time_index,                     # :128-131
                                    'Time Index': range(len(xarray_time_index))})
    result_df = pd.DataFrame()                                                         # :133
    for boundary, group_df in bc_df.groupby('RAS2D_TS_Name'):                          # :134
        merged_group = pd.merge_asof(model_dataframe, group_df, on='Datetime')         # :136-140
        merged_group['Concentration'] = merged_group['Concentration'].interpolate(method='linear')   # :142-144
        result_df = pd.concat([result_df, merged_group], ignore_index=True)            # :146-149
    boundary_df = pd.merge(result_df, flow_field_boundaries, left_on='RAS2D_TS_Name', right_on='Name', how='left')  # :152-158
    f2 = np.asarray(mesh[EDGES_FACE2])
    boundary_df['Ghost Cell'] = f2[boundary_df['Face Index'].to_list()]                # :159
    input_array[[boundary_df['Time Index']], [boundary_df['Ghost Cell']]] = boundary_df['Concentration']   # :163
    return input_array


class OracleModel:
    """transport.py:68-276 restated over arrays: owns the mesh dict, the LHS, the
    constituents and time_step; update() follows transport.py:201-276."""

    def __init__(self, mesh: dict, input_arrays: dict):
        if COEFFICIENT_TO_DIFFUSION_TERM not in mesh:
            derive_coefficients(mesh)
        self.mesh = mesh
        self.time_step = 0                                                   # :102
        self.lhs = LHS(mesh)                                                 # :152
        self.constituent_dict = {k: Constituent(k, mesh, v) for k, v in input_arrays.items()}
        self.last_A = None

    def update(self, update_concentration: dict | None = None):
        mesh = self.mesh
        t = self.time_step
        n = mesh[NUMBER_OF_REAL_CELLS] + 1
        self.lhs.update_values(mesh, t)                                      # :209-212
        A = self.lhs.csr()                                                   # :215-218
        self.last_A = A
        for name, con in self.constituent_dict.items():                      # :231
            if isinstance(update_concentration, dict) and name in update_concentration:
                con.state[t][0:n] = np.asarray(update_concentration[name])[0:n]   # :233-236
                x = np.asarray(update_concentration[name])[0:n]
            else:
                x = con.state[t][0:n]                                        # :238
            con.b.update_values(x, mesh, t)                                  # :241-246
            x = spsolve(A.tocsr(), con.b.vals)                               # :249
            con.state[t + 1][0:n] = x                                        # :252-257
            nz = np.nonzero(con.input_array[t + 1])[0]                       # :258
            con.state[t + 1][nz] = con.input_array[t + 1][nz]                # :259-264
            adv, dif, tot = mass_flux(mesh, con.state[t + 1], t)             # :267-273
            con.advection_mass_flux[t] = adv
            con.diffusion_mass_flux[t] = dif
            con.total_mass_flux[t] = tot
        self.time_step += 1                                                  # :276


# --------------------------------------------------------------------------- 8f-4: global mass balance
def mass_bal_global(model: 'OracleModel', constituent_name: str, lines) -> dict:
    """postproc_util.py:21-166 restated over arrays.  lines: [(name, face ids)] in 'BC Line ID' order (:72-74).
    dtype semantics kept: volumes are float32 and xarray sums them in float32 with NaN skipped (:37,41,94-95);
    total_mass_flux is a numpy array, so its sums propagate NaN (:99-100) and np.where(x <= 0, x, x * 0) (:118,136)
    keeps NaN in both parts."""
    mesh = model.mesh
    con = model.constituent_dict[constituent_name]
    n = mesh[NUMBER_OF_REAL_CELLS] + 1                                        # :35
    vol = np.asarray(mesh[VOLUME], dtype=np.float32)
    t_max = len(mesh['time_seconds']) - 1                                     # :49
    d = {'Vol_start': np.nansum(vol[0][0:n], dtype=np.float32),               # :36-41
         'Mass_start': (vol[0][0:n] * con.state[0][0:n]).sum(),               # :38-42 (float32 * float64)
         'Vol_end': np.nansum(vol[t_max][0:n], dtype=np.float32),             # :50-55
         'Mass_end': (vol[t_max][0:n] * con.state[t_max][0:n]).sum()}
    flow = np.asarray(mesh[FLOW_ACROSS_FACE], dtype=np.float32)
    dt = np.asarray(mesh[CHANGE_IN_TIME], dtype=np.float64)
    tot = dict.fromkeys(('v', 'vi', 'vo', 'm', 'mi', 'mo'), 0.0)
    for name, faces in lines:                                                 # :84
        faces = np.asarray(faces, dtype=np.int64)
        edge_vol = flow[:, faces] * dt[:, None]                               # :92-93 (promotes to float64)
        v = np.nansum(edge_vol)                                               # :94 (xarray: skipna)
        mass = con.total_mass_flux[:, faces]                                  # :99
        m = mass.sum()                                                        # :100 (numpy: NaN propagates)
        vi = np.nansum(np.where(edge_vol <= 0, edge_vol, 0.0))                # :108-109 (.where(cond, other=0))
        mi = np.where(mass <= 0, mass, mass * 0).sum()                        # :118-119
        vo = np.nansum(np.where(edge_vol >= 0, edge_vol, 0.0))                # :126-127
        mo = np.where(mass >= 0, mass, mass * 0).sum()                        # :136-137
        d[f'{name}_vol'], d[f'{name}_mass'] = v, m
        d[f'{name}_in_vol'], d[f'{name}_in_mass'] = vi, mi
        d[f'{name}_out_vol'], d[f'{name}_out_mass'] = vo, mo
        for key, val in zip(('v', 'vi', 'vo', 'm', 'mi', 'mo'), (v, vi, vo, m, mi, mo)):
            tot[key] = tot[key] + val
    d.update(bcTotalVolInOutAll=tot['v'], bcTotalVolInAll=tot['vi'], bcTotalVolOutAll=tot['vo'],     # :145-151
             bcTotalMassInOutAll=tot['m'], bcTotalMassInAll=tot['mi'], bcTotalMassOutAll=tot['mo'])
    with np.errstate(divide='ignore', invalid='ignore'):
        vol_end_calc = d['Vol_start'] + -1 * tot['vi'] + -1 * tot['vo']       # :153
        mass_end_calc = d['Mass_start'] + -1 * tot['mi'] + -1 * tot['mo']     # :154
        d['vol_end_calc'] = vol_end_calc                                      # :156
        d['error_vol'] = vol_end_calc - d['Vol_end']                          # :157
        d['prct_error_vol'] = np.float64(d['error_vol']) / np.float64(tot['vi']) * 100      # :158
        d['mass_end_calc'] = mass_end_calc                                    # :160
        d['error_mass'] = mass_end_calc - d['Mass_end']                       # :161
        d['prct_error_mass'] = np.float64(d['error_mass']) / np.float64(tot['mi']) * 100    # :162
    return d


# --------------------------------------------------------------------------- helpers for tests / bench
def parse_ras_stamps(stamps) -> np.ndarray:
    """'%d%b%Y %H:%M:%S' stamps (io/hdf.py:155-156) -> seconds since the first one."""
    from datetime import datetime
    ts = [datetime.strptime(str(s), '%d%b%Y %H:%M:%S') for s in stamps]
    return np.array([(x - ts[0]).total_seconds() for x in ts], dtype=np.float64)


def mesh_from_fixture(npz, diffusion_coefficient: float) -> dict:
    """Build the oracle's mesh dict from a tests/golden/*_inputs.npz file."""
    mesh = {
        EDGES_FACE1: npz['edges_face1'], EDGES_FACE2: npz['edges_face2'],
        NUMBER_OF_REAL_CELLS: int(npz['edges_face1'].max()),                 # io/hdf.py:268
        'face_x': npz['face_x'], 'face_y': npz['face_y'],
        FLOW_ACROSS_FACE: npz['face_flow'], EDGE_VELOCITY: npz['edge_velocity'],
        VOLUME: npz['volume'],
        'time_seconds': parse_ras_stamps(npz['time_stamps']),
        'diffusion_coefficient': diffusion_coefficient,
    }
    return derive_coefficients(mesh)
