"""HEC-RAS 2D HDF5 -> the engine's arrays, needs h5py.

Reads exactly the datasets the reference's reader reads
(/root/reference/src/clearwater_riverine/io/hdf.py:39-70 paths, :193-218 coordinates and time stamps,
:246-269 topology and nreal, :275-310 hydrodynamics, :355-436 boundary lines with the 'Faces'
attribute fix) and returns a ``model.Mesh`` keyed by the reference's variable names, with
``attrs['boundary_faces']`` = {boundary-line name: [face indices]} for the CSV boundary pipeline
(constituents.py:153-164).  The HDF files that lack 'Cell Volume' / 'Face Flow' (the reference's
numba fallback, utilities.py:463-512) are not supported: that branch is out of scope.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import numpy as np

from .model import Mesh

_BASE = 'Results/Unsteady/Output/Output Blocks/Base Output/Unsteady Time Series'


def _have_pandas() -> bool:
    try:
        import pandas  # noqa: F401
        return True
    except ImportError:
        return False


def boundary_dataframe(external_faces, attributes, keep):
    """mesh.attrs['boundary_data'] as the reference builds it (io/hdf.py:355-436): one row per kept boundary face with the
    columns of 'External Faces' (minus 'Station Start' / 'Station End') followed by the columns of the line 'Attributes'
    (byte strings decoded), which postproc_util.py:72-90 groups by 'Name' / 'BC Line ID' and reads 'Face Index' from."""
    import pandas as pd
    ext = pd.DataFrame(external_faces)
    att = pd.DataFrame(attributes)
    for col in att.select_dtypes([object]).columns:
        att[col] = att[col].str.decode('utf-8')
    att['BC Line ID'] = att.index
    df = pd.merge(ext, att, on='BC Line ID', how='left')[np.asarray(keep, dtype=bool)]
    df = df.drop(columns=[c for c in ('Station Start', 'Station End') if c in df.columns]).drop_duplicates()
    # the reference concatenates line by line in order of first appearance of the name (io/hdf.py:404-425)
    first = {nm: i for i, nm in enumerate(pd.unique(df['Name']))}
    return df.iloc[np.argsort(df['Name'].map(first).to_numpy(), kind='stable')]


def _parse_stamps(raw) -> np.ndarray:
    """'%d%b%Y %H:%M:%S' byte strings (io/hdf.py:152-156) -> datetime64[ns]."""
    from datetime import datetime
    out = [datetime.strptime(s.decode('utf8'), '%d%b%Y %H:%M:%S') for s in raw]
    return np.array(out, dtype='datetime64[ns]')


def _time_slice(stamps: np.ndarray, datetime_range) -> Tuple[Optional[int], Optional[int]]:
    """io/hdf.py:158-183: None, an inclusive (int, int) pair, or a pair of '%m-%d-%Y %H:%M:%S' strings."""
    if datetime_range is None:
        return None, None
    a, b = datetime_range
    if isinstance(a, (int, np.integer)):
        return int(a), int(b) + 1
    if isinstance(a, str):
        from datetime import datetime
        lo = np.datetime64(datetime.strptime(a, '%m-%d-%Y %H:%M:%S'))
        hi = np.datetime64(datetime.strptime(b, '%m-%d-%Y %H:%M:%S'))
        idx = np.nonzero((stamps >= lo) & (stamps <= hi))[0]
        return int(idx[0]), int(idx[-1]) + 1
    raise TypeError('Invalid datetime_range, must be tuple of strings or ints')          # io/hdf.py:180-183


def read_ras_hdf(file_path: str, datetime_range: Optional[Union[Tuple[int, int], Tuple[str, str]]] = None, lazy: bool = False) -> Mesh:
    """lazy=True (round 6): everything but the three per-level arrays ('Face Flow', 'Face Velocity', 'Cell Volume': the (T, .) bulk of the
    file) is read now; those are left in the file behind mesh.attrs['level_source'] (levels.HdfLevelSource: hyperslab reads by level, the
    same datetime_range window, io/hdf.py:149-191) for a run that streams them through the engine's ring."""
    try:
        import h5py
    except ImportError as exc:                                    # the image's default interpreter has no h5py
        raise ImportError('read_ras_hdf needs h5py; construct ClearwaterRiverine(mesh=...) from arrays instead') from exc
    with h5py.File(file_path, 'r') as f:
        area = f['Geometry/2D Flow Areas/Attributes'][()][0][0].decode('UTF-8')       # io/hdf.py:143-145
        geom = f[f'Geometry/2D Flow Areas/{area}']
        res = f[f'{_BASE}/2D Flow Areas/{area}']
        stamps = _parse_stamps(f[f'{_BASE}/Time Date Stamp'][()])
        t0, t1 = _time_slice(stamps, datetime_range)
        sl = slice(t0, t1)
        faces_cells = geom['Faces Cell Indexes'][()]
        centers = geom['Cells Center Coordinate'][()]
        if 'Cell Volume' not in res or 'Face Flow' not in res:
            raise NotImplementedError("HDF without 'Cell Volume' / 'Face Flow' optional outputs: the reference's "
                                      'fallback (utilities.py:463-512) is outside the transport path')
        m = Mesh({
            'edges_face1': faces_cells[:, 0].astype(np.int32),
            'edges_face2': faces_cells[:, 1].astype(np.int32),
            'face_x': centers[:, 0].astype(np.float64),
            'face_y': centers[:, 1].astype(np.float64),
            'time': stamps[sl],
        })
        if lazy:
            from .levels import HdfLevelSource
            first = 0 if t0 is None else t0
            m.attrs['level_source'] = HdfLevelSource(file_path, area, first, len(m['time']), int(faces_cells.shape[0]), int(centers.shape[0]))
        else:
            m['face_flow'] = res['Face Flow'][sl].astype(np.float32)
            m['edge_velocity'] = res['Face Velocity'][sl].astype(np.float32)
            m['volume'] = res['Cell Volume'][sl].astype(np.float32)
        m.attrs['nreal'] = int(faces_cells[:, 0].max())                                  # io/hdf.py:268-269
        # boundary lines (io/hdf.py:355-436): External Faces joined with the line attributes on 'BC Line ID', keeping of
        # every line only the faces listed in '<name> - Flow per Face'.attrs['Faces'] (the HEC-RAS bug the reference works
        # around), 'Station Start' / 'Station End' dropped, duplicates removed -- vectorised over all faces
        ext = f['Geometry/Boundary Condition Lines/External Faces'][()]
        attrs = f['Geometry/Boundary Condition Lines/Attributes'][()]
        names = [row[0].decode('utf-8') for row in attrs]
        line_of_face = np.asarray(ext['BC Line ID'], dtype=np.int64)
        face_index = np.asarray(ext['Face Index'], dtype=np.int64)
        keep = np.zeros(len(face_index), dtype=bool)
        for line_id, name in enumerate(names):
            fix = np.asarray(f[f'{_BASE}/Boundary Conditions/{name} - Flow per Face'].attrs['Faces'], dtype=np.int64)
            keep |= (line_of_face == line_id) & np.isin(face_index, fix)
        faces = {name: np.unique(face_index[keep & (line_of_face == line_id)]).tolist() for line_id, name in enumerate(names)}
        m.attrs['boundary_faces'] = faces
        m.attrs['boundary_data'] = boundary_dataframe(ext, attrs, keep) if _have_pandas() else faces
    return m
