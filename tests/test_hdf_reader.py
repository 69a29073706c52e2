"""The optional HDF reader against the golden extracts.

h5py lives only in the build container's conda interpreter, so the reader runs THERE in a child process (on the
reference's own HDF files) and dumps what it read; this test -- in the default interpreter, part of the default
`pytest -m "not gpu"` flow -- compares the dump with the committed extracts, the reference's fixture facts
(/root/reference/tests/test_riverine.py:78-86) and the reference's boundary DataFrame layout (io/hdf.py:355-436).
Skipped only where the reference files or that interpreter do not exist (the GPU box)."""
import os
import subprocess
import sys

import numpy as np
import pytest

REF = '/root/reference/tests/data/simple_test_cases'
H5PY_PYTHON = '/opt/conda/bin/python3.9'
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

try:
    import h5py  # noqa: F401
    _PY = sys.executable                             # an interpreter that has h5py itself: no child needed, same code path
except ImportError:
    _PY = H5PY_PYTHON

needs_files = pytest.mark.skipif(not (os.path.isdir(REF) and os.path.exists(_PY)),
                                 reason='reference HDF fixtures or an interpreter with h5py not present')

_DUMP = r"""
import sys, json, warnings
warnings.filterwarnings('ignore')
import numpy as np
sys.path.insert(0, sys.argv[1])
from clearwater_riverine_amd.hdf_reader import read_ras_hdf
path, out = sys.argv[2], sys.argv[3]
rng = json.loads(sys.argv[4])
m = read_ras_hdf(path, datetime_range=tuple(rng) if rng is not None else None)
bd = m.attrs['boundary_data']
cols = list(bd.columns) if hasattr(bd, 'columns') else []
np.savez(out, **{k: np.asarray(m[k]) for k in ('edges_face1', 'edges_face2', 'face_x', 'face_y', 'face_flow', 'edge_velocity', 'volume')},
         n_time=len(m['time']), nreal=m.attrs['nreal'],
         bd_columns=np.asarray(cols, dtype=str), bd_name=np.asarray(bd['Name'], dtype=str) if cols else np.asarray([], dtype=str),
         bd_face=np.asarray(bd['Face Index']) if cols else np.asarray([]), bd_line=np.asarray(bd['BC Line ID']) if cols else np.asarray([]),
         faces_json=json.dumps(m.attrs['boundary_faces']))
"""


def read_in_child(tmp_path, rel, datetime_range):
    import json
    out = str(tmp_path / 'dump.npz')
    subprocess.run([_PY, '-c', _DUMP, ROOT, os.path.join(REF, rel), out, json.dumps(datetime_range)], check=True,
                   capture_output=True, text=True)
    return np.load(out, allow_pickle=False)


@needs_files
@pytest.mark.parametrize('plan,rel,keep', [('plan01', 'plan01_10x5/clearWaterTestCases.p01.hdf', 64),
                                           ('plan02', 'plan02_2x1/clearWaterTestCases.p02.hdf', 25)])
def test_reader_matches_golden_extract(tmp_path, plan, rel, keep):
    import json
    z = np.load(os.path.join(GOLDEN, f'{plan}_inputs.npz'))
    m = read_in_child(tmp_path, rel, [0, keep - 1])
    for key in ('edges_face1', 'edges_face2', 'face_x', 'face_y', 'face_flow', 'edge_velocity', 'volume'):
        assert np.array_equal(m[key], z[key]), key
    assert int(m['n_time']) == keep and int(m['nreal']) == int(z['edges_face1'].max())
    faces = json.loads(str(m['faces_json']))
    names = [str(n) for n in z['bc_line_names']]
    for i, nm in enumerate(names):
        assert faces[nm] == [int(f) for f, l in zip(z['bc_face_index'], z['bc_face_line']) if l == i]
    # mesh.attrs['boundary_data'] in the reference's DataFrame layout (io/hdf.py:376-436): External Faces columns minus the
    # two station columns, then the line attributes; one row per kept face
    cols = [str(c) for c in m['bd_columns']]
    assert cols[:2] == ['BC Line ID', 'Face Index'] and 'Name' in cols and 'Station Start' not in cols and 'Station End' not in cols
    assert sorted(zip(m['bd_line'].tolist(), m['bd_face'].tolist())) == sorted(zip(z['bc_face_line'].tolist(), z['bc_face_index'].tolist()))
    assert [str(n) for n in m['bd_name']] == [names[int(l)] for l in m['bd_line']]


@needs_files
def test_datetime_range_counts_of_reference_test(tmp_path):
    """reference tests/test_riverine.py:78-86: 25 stamps; (5, 8) -> 4; the 12:00-12:10 string range -> 3."""
    rel = 'plan02_2x1/clearWaterTestCases.p02.hdf'
    assert int(read_in_child(tmp_path, rel, None)['n_time']) == 25
    assert int(read_in_child(tmp_path, rel, [5, 8])['n_time']) == 4
    assert int(read_in_child(tmp_path, rel, ['01-01-2023 12:00:00', '01-01-2023 12:10:00'])['n_time']) == 3
