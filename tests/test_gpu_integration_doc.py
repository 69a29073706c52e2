"""The reference-side binding of INTEGRATION.md section 2 -- the patch a maintainer would paste into
/root/reference/src/clearwater_riverine/transport.py (the __init__ lines after initialize_constituents, :152-156, and the
replacement body of update(), :201-276) -- EXECUTED, not grepped.

The code block is cut out of INTEGRATION.md as it stands and run against a duck-typed model: a Dataset stand-in with exactly the
surface the patch touches (m[name].values, m[name][t][0:n] = ..., m[name][t + 1] = ..., m.nreal, m.attrs, len(m.nface)) and
Constituent stand-ins with input_array and the three flux arrays (constituents.py:17-75).  xarray itself is not in this image; the
patch uses nothing of it beyond that surface.  Checked against the oracle (the reference's algorithm restated) and the facade.
"""
import os
import re

import numpy as np
import pytest

import cwr_oracle as oracle
from util import flux_err, load_plan, oracle_run, rel_err

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def integration_patch():
    """(init lines, update() source) of the first python block of INTEGRATION.md section 2."""
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = text[text.index('## 2. The patch to `transport.py`'):]
    block = re.search(r'```python\n(.*?)```', sec, re.S).group(1)
    init_src, upd_src = block.split('# --- replacement body of update()', 1)
    upd_src = upd_src[upd_src.index('def update('):]
    assert 'self._gpu.step(' in upd_src and 'TransportEngine(' in init_src
    return init_src, upd_src


class DataArrayLike:
    """The slice of xr.DataArray the patch uses: .values, integer / slice item access returning a view, item assignment."""

    def __init__(self, values):
        self.values = values

    def __getitem__(self, idx):
        return DataArrayLike(self.values[idx])

    def __setitem__(self, idx, value):
        self.values[idx] = getattr(value, 'values', value)


class DatasetLike:
    """The slice of xr.Dataset the patch uses: ds[name] -> DataArray over the SAME storage every time, ds.nreal, ds.attrs, ds.nface."""

    def __init__(self, arrays, attrs, ncell):
        self._arrays = arrays
        self.attrs = attrs
        self.nreal = attrs['nreal']
        self.nface = np.arange(ncell)                      # the 'nface' dimension: cells (io/hdf.py names cells "faces")

    def __getitem__(self, name):
        return DataArrayLike(self._arrays[name])


class ConstituentLike:
    def __init__(self, input_array, T, E):
        self.input_array = input_array
        self.advection_mass_flux = np.zeros((T, E))
        self.diffusion_mass_flux = np.zeros((T, E))
        self.total_mass_flux = np.zeros((T, E))


class ReferenceModelLike:
    """What ClearwaterRiverine.__init__ has built by transport.py:152-156: mesh, constituent_dict, time_step."""

    def __init__(self, mesh, inputs3):
        T, ncell, K = inputs3.shape
        E = len(mesh['edges_face1'])
        arrays = {k: np.asarray(v) for k, v in mesh.items() if hasattr(v, 'shape')}
        self.constituent_dict = {}
        for k in range(K):
            name = f'c{k}'
            st = np.full((T, ncell), np.nan)               # constituents.py:39-48
            st[0] = inputs3[0, :, k]                        # constituents.py:94-98
            arrays[name] = st
            self.constituent_dict[name] = ConstituentLike(np.ascontiguousarray(inputs3[:, :, k]), T, E)
        self.mesh = DatasetLike(arrays, {'nreal': int(mesh['nreal']), 'diffusion_coefficient': float(mesh['diffusion_coefficient'])}, ncell)
        self.time_step = 0


def bind(mesh, inputs3):
    init_src, upd_src = integration_patch()
    model = ReferenceModelLike(mesh, inputs3)
    ns = {'self': model, 'np': np}
    exec(compile(init_src, 'INTEGRATION.md section 2 (__init__)', 'exec'), ns)
    exec(compile(upd_src, 'INTEGRATION.md section 2 (update)', 'exec'), ns)
    return model, ns['update']


@pytest.mark.parametrize('plan,D,K', [('plan02', 0.01, 1), ('plan01', 0.01, 3)])
def test_the_pasted_update_body_runs_and_matches_oracle_and_facade(gpu_lib, plan, D, K):
    import clearwater_riverine_amd as cw
    from util import multi_inputs
    mesh, inp, _ = load_plan(plan, D)
    inputs3 = multi_inputs(inp, K, seed=5)
    steps = min(12, inputs3.shape[0] - 1)
    n = mesh['nreal'] + 1
    # an override at step 4, as the coupling loop passes it (transport.py:233-236): {name: DataArray}
    over_vals = 3.0 + np.arange(len(mesh['face_x']), dtype=np.float64) % 7
    overrides = {4: {'c0': over_vals}} if steps > 5 else {}
    ref = oracle_run(mesh, inputs3, steps, overrides=overrides)
    model, update = bind(mesh, inputs3)
    facade = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for s in range(steps):
        upd = overrides.get(s)
        update(model, {k: DataArrayLike(v) for k, v in upd.items()} if upd else None)
        facade.update(upd)
    assert model.time_step == steps
    for k in range(K):
        name = f'c{k}'
        want = ref.constituent_dict[name]
        got = model.mesh[name].values
        assert rel_err(got[:steps + 1], want.state[:steps + 1]) <= 1e-9          # NaN ghost pattern included
        assert rel_err(got[:steps + 1], facade.mesh[name][:steps + 1]) <= 1e-9
        for arr in ('advection_mass_flux', 'diffusion_mass_flux', 'total_mass_flux'):
            assert flux_err(getattr(model.constituent_dict[name], arr)[:steps], getattr(want, arr)[:steps]) <= 1e-8
    model._gpu.close()


def test_the_pasted_binding_on_a_mesh_large_enough_to_be_renumbered(gpu_lib):
    """n > 4 096: the patch's lane_order(...) line runs (internal numbering; ids at the boundary stay the mesh's)."""
    import clearwater_riverine_amd as cw
    K, steps = 2, 3
    mesh = cw.synthetic.make_mesh(90, 60, steps, seed=31, n_merge=120, dt=40.0, diffusion_coefficient=0.5)
    oracle.derive_coefficients(mesh)
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=31)
    assert mesh['nreal'] + 1 > 4096
    ref = oracle_run(mesh, inputs3, steps)
    model, update = bind(mesh, inputs3)
    assert model._gpu.state_row_order() is not None
    for _ in range(steps):
        update(model)
    for k in range(K):
        assert rel_err(model.mesh[f'c{k}'].values[:steps + 1], ref.constituent_dict[f'c{k}'].state[:steps + 1]) <= 1e-9
    model._gpu.close()


def streaming_patch():
    """The python block of INTEGRATION.md section 2a (streaming a long file), with its one file-bound line -- `src = HdfLevelSource(...)` --
    replaced by a source over the Dataset's own arrays (h5py is not on the GPU box; levels.HdfLevelSource is tested against the reference's HDF
    in tests/test_levels.py)."""
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = text[text.index('## 2a. Streaming a long file'):]
    block = re.search(r'```python\n(.*?)```', sec, re.S).group(1)
    assert 'src = HdfLevelSource(' in block and 'FlowWindowFeeder(' in block and 'self._feed.fill(0)' in block
    block = re.sub(r'^src = HdfLevelSource\(.*$', "src = ArrayLevelSource(m['face_flow'].values, m['edge_velocity'].values, m['volume'].values)", block, flags=re.M)
    return block


@pytest.mark.parametrize('case', ['plan01', 'renumbered'])
def test_the_pasted_streaming_binding_equals_the_resident_binding_bit_for_bit(gpu_lib, case):
    """INTEGRATION.md section 2a executed: after section 2's lines, the flow field is re-opened as a ring of 16 levels fed from a level source,
    the boundary values travel with the levels (cwr_load_boundary(NULL) + cwr_boundary_window_load), `self._feed.fill(t)` precedes every step.
    The histories equal those of the resident binding of section 2 BIT FOR BIT (deterministic passes: K <= 8)."""
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.levels import ArrayLevelSource
    from util import multi_inputs
    if case == 'plan01':
        mesh, inp, _ = load_plan('plan01', 0.01)
        inputs3 = multi_inputs(inp, 3, seed=5)
        steps = 40                                                # 64 levels through W = 16
    else:
        steps = 24
        mesh = cw.synthetic.make_mesh(90, 60, steps, seed=31, n_merge=120, dt=40.0, diffusion_coefficient=0.5)
        oracle.derive_coefficients(mesh)
        inputs3 = cw.synthetic.distinct_input_array(mesh, 2, seed=31)
    K = inputs3.shape[2]
    resident, update = bind(mesh, inputs3)
    for _ in range(steps):
        update(resident)
    streamed, update2 = bind(mesh, inputs3)
    ns = {'self': streamed, 'np': np, 'm': streamed.mesh, 'names': streamed._names, 'n': int(mesh['nreal']) + 1,
          'order': streamed._gpu._order, 'ArrayLevelSource': ArrayLevelSource}
    exec(compile(streaming_patch(), 'INTEGRATION.md section 2a', 'exec'), ns)
    assert streamed._feed.W == 16 or streamed._feed.W == inputs3.shape[0]
    for _ in range(steps):
        streamed._feed.fill(streamed.time_step)                   # ("in update(), before self._gpu.step")
        update2(streamed)
    for k in range(K):
        assert np.array_equal(resident.mesh[f'c{k}'].values, streamed.mesh[f'c{k}'].values, equal_nan=True)
        assert np.array_equal(resident.constituent_dict[f'c{k}'].total_mass_flux, streamed.constituent_dict[f'c{k}'].total_mass_flux, equal_nan=True)
    streamed._feed.close()
    resident._gpu.close(); streamed._gpu.close()
