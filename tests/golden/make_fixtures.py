#!/opt/conda/bin/python3.9
"""Extract small, time-windowed golden INPUT fixtures from the HEC-RAS 2D HDF5
files the reference's own tests use.

Run in the build container only (needs h5py, which lives in
/opt/conda/bin/python3.9 there):

    /opt/conda/bin/python3.9 tests/golden/make_fixtures.py

Source files (data, not code) under /root/reference/tests/data/simple_test_cases:
    plan01_10x5/clearWaterTestCases.p01.hdf  (50 real + 30 ghost cells, 115 faces, dt = 1 s)
    plan02_2x1/clearWaterTestCases.p02.hdf   (2 real + 6 ghost cells, 7 faces, dt = 300 s)
    plan03_2x1/clearWaterTestCases.p03.hdf   (same mesh, dt = 1 s)

The HDF dataset paths read here are the ones the reference's reader uses
(/root/reference/src/clearwater_riverine/io/hdf.py:39-70, topology :246-269,
hydrodynamics :275-310, boundary lines + the 'Faces' attribute fix :355-436).
Only arrays are written (np.savez_compressed); no reference source travels.

Expected OUTPUTS for these inputs are produced by tests/golden/make_expected.py
from the CPU oracle (oracle/cwr_oracle.py); the reference package itself cannot
be imported in this image (xarray/holoviews/geoviews/geopandas absent).
"""
import os
import sys

import h5py
import numpy as np

REF = '/root/reference/tests/data/simple_test_cases'
HERE = os.path.dirname(os.path.abspath(__file__))

PLANS = {
    # name: (relative hdf path, number of time stamps kept, IC csv, BC csv)
    'plan01': ('plan01_10x5/clearWaterTestCases.p01.hdf', 64,
               'plan01_10x5/cwr_initial_conditions_p01.csv',
               'plan01_10x5/cwr_boundary_conditions_p01.csv'),
    'plan02': ('plan02_2x1/clearWaterTestCases.p02.hdf', 25,
               'plan02_2x1/cwr_initial_conditions_p02.csv',
               'plan02_2x1/cwr_boundary_conditions_p02.csv'),
    'plan03': ('plan03_2x1/clearWaterTestCases.p03.hdf', 64,
               'plan03_2x1/cwr_initial_conditions_p03.csv',
               'plan03_2x1/cwr_boundary_conditions_p03.csv'),
}

BASE = 'Results/Unsteady/Output/Output Blocks/Base Output/Unsteady Time Series'


def read_csv_rows(path, max_rows_per_key=None):
    """Tiny CSV reader (no pandas in this interpreter's contract): returns header, rows."""
    with open(path) as fh:
        lines = [ln.strip() for ln in fh.read().splitlines()]
    header = lines[0].split(',')
    rows = []
    seen = {}
    for ln in lines[1:]:
        parts = ln.split(',')
        if len(parts) < len(header) or all(p == '' for p in parts):
            continue  # blank padding rows (plan01's BC csv has 1368 of them)
        key = parts[0]
        seen[key] = seen.get(key, 0) + 1
        if max_rows_per_key is not None and seen[key] > max_rows_per_key:
            continue
        rows.append(parts[:len(header)])
    return header, rows


def extract(name, hdf_rel, nkeep, ic_rel, bc_rel):
    f = h5py.File(os.path.join(REF, hdf_rel), 'r')
    area = f['Geometry/2D Flow Areas/Attributes'][()][0][0].decode('UTF-8')
    geom = f[f'Geometry/2D Flow Areas/{area}']
    res = f[f'{BASE}/2D Flow Areas/{area}']

    faces_cells = geom['Faces Cell Indexes'][()]            # (E, 2) int32
    centers = geom['Cells Center Coordinate'][()]          # (ncell, 2) float64
    stamps_all = f[f'{BASE}/Time Date Stamp'][()]
    stamps = np.array([s.decode('utf8') for s in stamps_all[:nkeep]])

    out = {
        'edges_face1': faces_cells[:, 0].astype(np.int32),
        'edges_face2': faces_cells[:, 1].astype(np.int32),
        'face_x': centers[:, 0].astype(np.float64),
        'face_y': centers[:, 1].astype(np.float64),
        'time_stamps': stamps,                                  # '%d%b%Y %H:%M:%S'
        'n_time_stamps_in_hdf': np.int64(len(stamps_all)),
        'face_flow': res['Face Flow'][:nkeep].astype(np.float32),
        'edge_velocity': res['Face Velocity'][:nkeep].astype(np.float32),
        'volume': res['Cell Volume'][:nkeep].astype(np.float32),
    }
    # the very last stored volume row: needed for the notebook known answer
    # "Mass_end = sum(V[-1, real] * 100)" (examples/dev_sandbox/test_functions_for_pytest.ipynb cell[1])
    out['volume_last'] = res['Cell Volume'][len(stamps_all) - 1].astype(np.float32)

    # boundary-condition lines -> face lists, with the reader's 'Faces' attribute fix
    ext = f['Geometry/Boundary Condition Lines/External Faces'][()]
    attrs = f['Geometry/Boundary Condition Lines/Attributes'][()]
    bc_names, bc_face_idx, bc_line_of_face = [], [], []
    for line_id, row in enumerate(attrs):
        bname = row[0].decode('utf-8')
        orig = [int(r['Face Index']) for r in ext if int(r['BC Line ID']) == line_id]
        fix = [int(v) for v in f[f'{BASE}/Boundary Conditions/{bname} - Flow per Face'].attrs['Faces']]
        kept = sorted(set(x for x in orig if x in fix))
        bc_names.append(bname)
        for fi in kept:
            bc_face_idx.append(fi)
            bc_line_of_face.append(line_id)
    out['bc_line_names'] = np.array(bc_names)
    out['bc_face_index'] = np.array(bc_face_idx, dtype=np.int32)
    out['bc_face_line'] = np.array(bc_line_of_face, dtype=np.int32)

    # IC / BC CSVs (data files of the reference's tests), trimmed to the kept window
    h, rows = read_csv_rows(os.path.join(REF, ic_rel))
    assert h == ['Cell_Index', 'Concentration'], h
    out['ic_cell_index'] = np.array([int(float(r[0])) for r in rows], dtype=np.int64)
    out['ic_concentration'] = np.array([float(r[1]) for r in rows], dtype=np.float64)
    h, rows = read_csv_rows(os.path.join(REF, bc_rel), max_rows_per_key=nkeep)
    assert h == ['RAS2D_TS_Name', 'Datetime', 'Concentration'], h
    out['bc_csv_name'] = np.array([r[0] for r in rows])
    out['bc_csv_datetime'] = np.array([r[1] for r in rows])        # '%m/%d/%Y %H:%M'
    out['bc_csv_concentration'] = np.array([float(r[2]) for r in rows], dtype=np.float64)

    path = os.path.join(HERE, f'{name}_inputs.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: E={faces_cells.shape[0]} ncell={centers.shape[0]} '
          f'nreal={faces_cells[:, 0].max()} T_kept={nkeep}/{len(stamps_all)} '
          f'bc={dict(zip(bc_names, [list(np.array(bc_face_idx)[np.array(bc_line_of_face) == i]) for i in range(len(bc_names))]))} '
          f'-> {os.path.getsize(path)} bytes')


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit('reference fixtures not present (this script only runs in the build container)')
    for k, v in PLANS.items():
        extract(k, *v)
