"""CPU tests of the output side: the zarr-v2 stream writer and the mass-balance host logic
(clearwater_riverine_amd.mass_balance) against the oracle's restatement of postproc_util._mass_bal_global."""
import json
import os

import numpy as np
import pytest

import cwr_oracle as oracle
from util import load_plan, multi_inputs, oracle_run


def test_zarr_stream_writer_layout_and_round_trip(tmp_path):
    from clearwater_riverine_amd.outputs import ZarrStreamWriter, read_zarr_level
    store = str(tmp_path / 'run.zarr')
    w = ZarrStreamWriter(store, {'salinity': (7, 'nface'), 'salinity_total_mass_flux': (5, 'nedge')}, 4, {'units': 'mg/L'})
    rng = np.random.default_rng(0)
    rows = {t: rng.standard_normal(7) for t in (0, 2, 3)}
    rows[2][3] = np.nan
    for t, r in rows.items():
        w.write_level('salinity', t, r)
    # metadata: what zarr v2 / xarray.open_zarr expect (io/outputs.py:11-17 writes consolidated=True)
    za = json.load(open(os.path.join(store, 'salinity', '.zarray')))
    assert za == {'zarr_format': 2, 'shape': [4, 7], 'chunks': [1, 7], 'dtype': '<f8', 'compressor': None,
                  'fill_value': 'NaN', 'order': 'C', 'filters': None}
    assert json.load(open(os.path.join(store, 'salinity', '.zattrs'))) == {'_ARRAY_DIMENSIONS': ['time', 'nface']}
    assert json.load(open(os.path.join(store, '.zgroup'))) == {'zarr_format': 2}
    cons = json.load(open(os.path.join(store, '.zmetadata')))
    assert cons['zarr_consolidated_format'] == 1 and cons['metadata']['salinity/.zarray'] == za
    assert cons['metadata']['.zattrs'] == {'units': 'mg/L'}
    assert cons['metadata']['salinity_total_mass_flux/.zattrs'] == {'_ARRAY_DIMENSIONS': ['time', 'nedge']}
    # chunks: raw little-endian float64, key "<t>.0"; an unwritten chunk reads as the fill value
    assert os.path.getsize(os.path.join(store, 'salinity', '2.0')) == 7 * 8
    for t, r in rows.items():
        assert np.array_equal(read_zarr_level(store, 'salinity', t), r, equal_nan=True)
    assert np.isnan(read_zarr_level(store, 'salinity', 1)).all()
    with pytest.raises(ValueError):
        w.write_level('salinity', 0, np.zeros(6))
    with pytest.raises(IndexError):
        w.write_level('salinity', 4, np.zeros(7))
    with pytest.raises(FileNotFoundError):                         # io/outputs.py:33-38
        ZarrStreamWriter(str(tmp_path / 'missing_dir' / 'x.zarr'), {'a': (1, 'nface')}, 1)


@pytest.mark.parametrize('plan,D', [('plan02', 0.01), ('plan01', 0.01)])
def test_mass_balance_host_logic_matches_the_restated_reference(plan, D):
    """assemble() over a ledger accumulated step by step == oracle.mass_bal_global over the full histories."""
    from clearwater_riverine_amd.mass_balance import assemble, boundary_lines, volume_columns
    mesh, inp, z = load_plan(plan, D)
    inputs3 = multi_inputs(inp, 2)
    steps = 24 if plan == 'plan02' else 30
    # cut the flow field to the simulated window so that "the end" is the last simulated level
    for key in ('face_flow', 'edge_velocity', 'volume', 'time_seconds', 'advection_coeff', 'coeff_to_diffusion',
                'edge_vertical_area', 'dt'):
        if key in mesh:
            mesh[key] = mesh[key][:steps + 1].copy()
    if 'dt' in mesh:
        mesh['dt'][-1] = np.nan                                   # utilities.py:537-541: trailing NaN
    inputs3 = inputs3[:steps + 1]
    model = oracle_run(mesh, inputs3, steps)
    faces = np.asarray(z['bc_face_index'])
    lines = boundary_lines({'US_Flow': faces[: max(1, len(faces) // 2)], 'DS_Stage': faces[max(1, len(faces) // 2):]})
    n = model.mesh['nreal'] + 1
    for k in range(2):
        want = oracle.mass_bal_global(model, f'c{k}', lines)
        con = model.constituent_dict[f'c{k}']
        ledger = np.zeros((len(lines), 3))
        for t in range(steps):                                    # what the device ledger accumulates, step by step
            for li, (_, f) in enumerate(lines):
                x = con.total_mass_flux[t, f]
                ledger[li] += [x.sum(), np.where(x <= 0, x, x * 0).sum(), np.where(x >= 0, x, x * 0).sum()]
        vol = model.mesh['volume'].astype(np.float64)
        got = assemble(lines, volume_columns(model.mesh['face_flow'], model.mesh['dt'], lines), ledger,
                       vol[0, :n].sum(), (vol[0, :n] * con.state[0, :n]).sum(),
                       vol[steps, :n].sum(), (vol[steps, :n] * con.state[steps, :n]).sum())
        assert list(got) == list(want)                             # same columns, same order as the reference's DataFrame
        for key in want:
            tol = 1e-6 if 'ol' in key.lower() and 'mass' not in key.lower() else 1e-10   # volumes: the reference sums float32
            if 'error' in key or 'prct' in key:
                continue                                           # differences of nearly equal sums: checked through their parts
            assert np.allclose(got[key], want[key], rtol=tol, atol=0, equal_nan=True), key
