"""The optional HDF reader against the golden extracts (needs h5py and the reference's HDF files: runs only in
the build container, with an interpreter that has h5py; skipped elsewhere)."""
import os

import numpy as np
import pytest

h5py = pytest.importorskip('h5py')
REF = '/root/reference/tests/data/simple_test_cases'
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference fixtures not present')
@pytest.mark.parametrize('plan,rel,keep', [('plan01', 'plan01_10x5/clearWaterTestCases.p01.hdf', 64),
                                           ('plan02', 'plan02_2x1/clearWaterTestCases.p02.hdf', 25)])
def test_reader_matches_golden_extract(plan, rel, keep):
    from clearwater_riverine_amd.hdf_reader import read_ras_hdf
    z = np.load(os.path.join(GOLDEN, f'{plan}_inputs.npz'))
    m = read_ras_hdf(os.path.join(REF, rel), datetime_range=(0, keep - 1))
    for key in ('edges_face1', 'edges_face2', 'face_x', 'face_y', 'face_flow', 'edge_velocity', 'volume'):
        assert np.array_equal(m[key], z[key]), key
    assert len(m['time']) == keep and m.attrs['nreal'] == int(z['edges_face1'].max())
    names = list(z['bc_line_names'])
    for i, nm in enumerate(names):
        assert m.attrs['boundary_faces'][nm] == [int(f) for f, l in zip(z['bc_face_index'], z['bc_face_line']) if l == i]


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference fixtures not present')
def test_datetime_range_counts_of_reference_test():
    """reference tests/test_riverine.py:78-86: 25 stamps; (5, 8) -> 4; the 12:00-12:10 string range -> 3."""
    from clearwater_riverine_amd.hdf_reader import read_ras_hdf
    p = os.path.join(REF, 'plan02_2x1/clearWaterTestCases.p02.hdf')
    assert len(read_ras_hdf(p)['time']) == 25
    assert len(read_ras_hdf(p, datetime_range=(5, 8))['time']) == 4
    assert len(read_ras_hdf(p, datetime_range=('01-01-2023 12:00:00', '01-01-2023 12:10:00'))['time']) == 3
