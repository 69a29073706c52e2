"""GPU tests of the output side, through the C ABI: the device mass-balance ledger
(cwr_set_boundary_lines / CWR_STEP_MASS_BALANCE / cwr_domain_mass) against the oracle's restatement of
postproc_util._mass_bal_global, and the streamed zarr output (cwr_output_*) against the RAM-resident history."""
import json
import os

import numpy as np
import pytest

import cwr_oracle as oracle
from util import load_plan, multi_inputs, oracle_run, rel_err
from test_gpu_parity import synthetic_case

pytestmark = pytest.mark.gpu


def ghost_face_lines(mesh, n_lines=3):
    """Boundary-condition lines of a synthetic mesh: its ghost faces, dealt into n_lines groups."""
    n = mesh['nreal'] + 1
    gf = np.nonzero(np.asarray(mesh['edges_face2']) >= n)[0]
    return {f'line{li}': gf[li::n_lines] for li in range(n_lines)}


def close_enough(got, want, key):
    if 'error' in key or 'prct' in key:
        return True                                # differences of nearly equal sums: checked through their parts
    is_vol = 'vol' in key.lower() and 'mass' not in key.lower()
    return np.allclose(got, want, rtol=1e-6 if is_vol else 1e-9, atol=0.0 if not is_vol else 1e-3, equal_nan=True)


@pytest.mark.parametrize('case', ['plan02', 'plan01', 'synthetic', 'synthetic-hilbert'])
def test_device_mass_balance_matches_restated_reference(gpu_lib, case, monkeypatch):
    import clearwater_riverine_amd as cw
    K = 3
    if case.startswith('plan'):
        mesh, inp, z = load_plan(case, 0.01)
        steps = 24 if case == 'plan02' else 20
        inputs3 = multi_inputs(inp, K)
        faces = np.asarray(z['bc_face_index'])
        lines = {'US_Flow': faces[: max(1, len(faces) // 2)], 'DS_Stage': faces[max(1, len(faces) // 2):]}
    else:
        monkeypatch.setenv('CWR_NO_SMALL', '1')
        mesh, inputs3 = synthetic_case(K, nx=40, ny=21, n_steps=10, seed=31, n_merge=30, n_dry=2)
        steps = 10
        lines = ghost_face_lines(mesh)
    for key in ('face_flow', 'edge_velocity', 'volume', 'time_seconds', 'advection_coeff', 'coeff_to_diffusion',
                'edge_vertical_area', 'dt'):
        if key in mesh:
            mesh[key] = np.asarray(mesh[key])[:steps + 1].copy()
    mesh['dt'][-1] = np.nan
    inputs3 = inputs3[:steps + 1]
    names = [f'c{k}' for k in range(K)]
    ref = oracle_run(mesh, inputs3, steps)
    m = dict(mesh)
    model = cw.ClearwaterRiverine(mesh=m, diffusion_coefficient_input=mesh['diffusion_coefficient'],
                                  input_arrays={nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)},
                                  store_history=False)
    if case == 'synthetic-hilbert':                                   # force the internal renumbering on a small mesh
        from clearwater_riverine_amd.ordering import hilbert_order
        n = mesh['nreal'] + 1
        model.engine.close()
        model.engine = cw.TransportEngine(mesh['edges_face1'], mesh['edges_face2'], len(mesh['face_x']), K,
                                          cell_order=hilbert_order(mesh['face_x'], mesh['face_y'], n))
        model.engine.load_flow_field(mesh['face_flow'], mesh['edge_velocity'], mesh['volume'], mesh['dt'],
                                     mesh['face_to_face_dist'], mesh['diffusion_coefficient'])
        model.engine.load_boundary(inputs3[:, n:, :])
    from clearwater_riverine_amd.mass_balance import boundary_lines
    model._lines = boundary_lines(lines)
    model.engine.set_boundary_lines([f for _, f in model._lines])
    for _ in range(steps):
        model.update()
    for nm in names:
        want = oracle.mass_bal_global(ref, nm, model._lines)
        got = model.mass_bal_global(nm)
        assert list(got) == list(want)
        for key in want:
            assert close_enough(got[key], want[key], key), (key, got[key], want[key])
    # lines with a face whose ghost cell has no boundary value: NaN in the reference, NaN here
    if case == 'synthetic':
        assert any(np.isnan(model.mass_bal_global('c0')[f'{name}_mass']) == np.isnan(oracle.mass_bal_global(ref, 'c0', model._lines)[f'{name}_mass'])
                   for name, _ in model._lines)


def test_mass_balance_errors(gpu_lib):
    import clearwater_riverine_amd as cw
    mesh, inputs3 = synthetic_case(2, nx=12, ny=8, n_steps=3, seed=5)
    n = mesh['nreal'] + 1
    eng = cw.TransportEngine(mesh['edges_face1'], mesh['edges_face2'], len(mesh['face_x']), 2)
    eng.load_flow_field(mesh['face_flow'], mesh['edge_velocity'], mesh['volume'], mesh['dt'], mesh['face_to_face_dist'], 0.1)
    eng.load_boundary(inputs3[:, n:, :])
    eng.set_state(inputs3[0, :n, :])
    with pytest.raises(IndexError):
        eng.step(0, mass_balance=True)                     # no lines registered
    with pytest.raises(ValueError):
        eng.set_boundary_lines([[len(mesh['edges_face1'])]])   # face id out of range
    eng.set_boundary_lines([[0, 1], []])
    eng.step(0, mass_balance=True)
    first = eng.get_mass_balance()
    assert first.shape == (2, 3, 2) and np.all(first[1] == 0.0)
    assert np.allclose(first[0, 0], first[0, 1] + first[0, 2], rtol=1e-12, atol=1e-300)
    eng.reset_mass_balance()
    assert np.all(eng.get_mass_balance() == 0.0)
    mass, vol = eng.domain_mass(1)
    assert vol == pytest.approx(float(np.asarray(mesh['volume'], np.float64)[1, :n].sum()), rel=1e-12)


@pytest.mark.parametrize('with_flux', [False, True])
@pytest.mark.parametrize('renumbered', [False, True])
def test_streamed_zarr_output_equals_ram_history(gpu_lib, tmp_path, with_flux, renumbered, monkeypatch):
    """output_store=...: every level written by the pinned-ring writer thread equals the reference-style RAM history
    of an identical run (bitwise), with a ring shorter than the run and no per-step state download."""
    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.outputs import read_zarr_level
    monkeypatch.setenv('CWR_NO_SMALL', '1')
    K, steps = 4, 9
    mesh, inputs3 = synthetic_case(K, nx=36, ny=17, n_steps=steps, seed=41, n_merge=20, n_dry=1)
    names = [f'c{k}' for k in range(K)]
    arrays = {nm: inputs3[:, :, k].copy() for k, nm in enumerate(names)}
    ram = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={k: v.copy() for k, v in arrays.items()})
    store = str(tmp_path / 'run.zarr')
    streamed = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={k: v.copy() for k, v in arrays.items()},
                                     store_history=False, host_state=False, output_store=store, output_flux=with_flux)
    if renumbered:
        pytest.importorskip('numpy')
        from clearwater_riverine_amd.ordering import hilbert_order
        n = mesh['nreal'] + 1
        for mdl in (ram, streamed):
            if mdl._stream is not None:
                mdl._stream.close()
            mdl.engine.close()
            mdl.engine = cw.TransportEngine(mesh['edges_face1'], mesh['edges_face2'], len(mesh['face_x']), K,
                                            cell_order=hilbert_order(mesh['face_x'], mesh['face_y'], n))
            mdl.engine.load_flow_field(mesh['face_flow'], mesh['edge_velocity'], mesh['volume'], mesh['dt'],
                                       mesh['face_to_face_dist'], mesh['diffusion_coefficient'])
            mdl.engine.load_boundary(inputs3[:, n:, :])
        from clearwater_riverine_amd.outputs import StreamedOutput
        streamed._stream = StreamedOutput(streamed.engine, store, names, steps + 1, with_flux=with_flux, n_slots=2)
    for _ in range(steps):
        ram.update()
        streamed.update()
    streamed.finalize()
    assert streamed._stream is None
    cons = json.load(open(os.path.join(store, '.zmetadata')))
    assert cons['metadata'][f'{names[0]}/.zarray']['shape'] == [steps + 1, len(mesh['face_x'])]
    for nm in names:
        for t in range(steps + 1):
            assert np.array_equal(read_zarr_level(store, nm, t), ram.mesh[nm][t], equal_nan=True), (nm, t)
        if with_flux:
            con = ram.constituent_dict[nm]
            for t in range(steps):
                assert np.array_equal(read_zarr_level(store, f'{nm}_total_mass_flux', t), con.total_mass_flux[t], equal_nan=True)
                assert np.array_equal(read_zarr_level(store, f'{nm}_advection_mass_flux', t), con.advection_mass_flux[t], equal_nan=True)
                assert np.array_equal(read_zarr_level(store, f'{nm}_diffusion_mass_flux', t), con.diffusion_mass_flux[t], equal_nan=True)
    # finalize(save=True, '*.zarr') of the RAM model writes the same store from its history (io/outputs.py:11-17)
    store2 = str(tmp_path / 'final.zarr')
    ram.finalize(save=True, output_filepath=store2)
    assert np.array_equal(read_zarr_level(store2, names[1], steps), read_zarr_level(store, names[1], steps), equal_nan=True)


def test_output_ring_api_errors(gpu_lib):
    import clearwater_riverine_amd as cw
    mesh, inputs3 = synthetic_case(2, nx=12, ny=8, n_steps=3, seed=5)
    n = mesh['nreal'] + 1
    eng = cw.TransportEngine(mesh['edges_face1'], mesh['edges_face2'], len(mesh['face_x']), 2)
    eng.load_flow_field(mesh['face_flow'], mesh['edge_velocity'], mesh['volume'], mesh['dt'], mesh['face_to_face_dist'], 0.1)
    eng.load_boundary(inputs3[:, n:, :])
    eng.set_state(inputs3[0, :n, :])
    with pytest.raises(IndexError):
        eng.output_push()                                  # not open
    eng.output_open(n_slots=2, with_flux=True)
    with pytest.raises(IndexError):
        eng.output_open()                                  # already open
    eng.step(0, mass_flux=False)
    with pytest.raises(IndexError):
        eng.output_push()                                  # fluxes requested but the step did not compute them
    eng.step(1, mass_flux=True)
    s = eng.output_push()
    state, flux = eng.output_wait(s)
    want = eng.get_state()
    assert np.array_equal(state.T, want, equal_nan=True)
    assert np.array_equal(flux[2].T, eng.get_mass_flux()[2], equal_nan=True)
    eng.output_release(s)
    with pytest.raises(IndexError):
        eng.output_wait(s)                                 # released: holds no snapshot
    eng.output_close()
    eng.close()


def test_snapshots_written_in_place_equal_those_of_the_copy_engine(gpu_lib, monkeypatch, capfd):
    """Round 5: snapshots of up to CWR_OUTPUT_DIRECT_MB (8 since round 6) are written by the snapshot kernels straight into the facade's
    page-locked history rows (no staging buffer, no copy command); larger ones -- or CWR_OUTPUT_DIRECT_MB=0 -- go through the copy
    engine.  Same histories bit for bit, fluxes included (total = advection + diffusion formed in the one flux launch), and the
    debug counters say which way each facade's snapshots went."""
    import clearwater_riverine_amd as cw
    K, steps = 3, 9
    mesh = cw.synthetic.make_mesh(60, 24, steps, seed=23, n_merge=40, dt=40.0, diffusion_coefficient=0.5)
    inputs3 = cw.synthetic.distinct_input_array(mesh, K, seed=23)
    monkeypatch.setenv('CWR_OUTPUT_DEBUG', '1')
    hist = {}
    for label, mb in (('in place', None), ('copy engine', '0')):
        if mb is None:
            monkeypatch.delenv('CWR_OUTPUT_DIRECT_MB', raising=False)
        else:
            monkeypatch.setenv('CWR_OUTPUT_DIRECT_MB', mb)
        mdl = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={f'c{k}': inputs3[:, :, k].copy() for k in range(K)}, deterministic=True)
        for _ in range(steps):
            mdl.update()
        hist[label] = ({nm: np.array(mdl.mesh[nm]) for nm in mdl.constituents},
                       {nm: [np.array(getattr(mdl.constituent_dict[nm], a)) for a in ('advection_mass_flux', 'diffusion_mass_flux', 'total_mass_flux')]
                        for nm in mdl.constituents})
        capfd.readouterr()
        mdl.close_output(); mdl.engine.close()
        err = capfd.readouterr().err
        want = f'{steps} snapshots written in place, 0 through' if mb is None else f'0 snapshots written in place, {steps} through'
        assert want in err, err
    for nm in hist['in place'][0]:
        assert np.array_equal(hist['in place'][0][nm], hist['copy engine'][0][nm], equal_nan=True)
        for a, b in zip(hist['in place'][1][nm], hist['copy engine'][1][nm]):
            assert np.array_equal(a, b, equal_nan=True)
        adv, dif, tot = hist['in place'][1][nm]
        assert np.array_equal(tot[:steps], adv[:steps] + dif[:steps], equal_nan=True)


def test_facade_histories_beside_a_busy_chip_equal_those_of_a_quiet_run(gpu_lib):
    """The facade's per-step read-out (snapshots written in place into page-locked rows, completion by an event) and the engine's
    copy-free convergence check (results noted to page-locked memory behind a sequence word), with another engine keeping the CUs and
    the copy engines busy from a second thread: the same histories bit for bit as alone."""
    import threading
    import clearwater_riverine_amd as cw
    from test_gpu_parity import make_engine
    K, steps = 3, 25
    mesh = cw.synthetic.make_mesh(109, 28, steps, seed=20100529 % 1000, n_merge=109, dx=75.0, dy=75.0, dt=3600.0, velocity=0.4, diffusion_coefficient=0.1,
                                  period_steps=24)
    inputs3 = cw.synthetic.boundary_input_array(mesh, K, inlet_period_s=24 * 3600.0)
    other = cw.synthetic.make_mesh(300, 300, 4, seed=9, n_merge=2000, dt=40.0, diffusion_coefficient=0.5)
    oracle.derive_coefficients(other)
    other_in = cw.synthetic.distinct_input_array(other, 8, seed=9)

    def history():
        mdl = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
        for _ in range(steps):
            mdl.update()
        out = ([np.array(mdl.mesh[nm]) for nm in mdl.constituents],
               [np.array(mdl.constituent_dict[nm].total_mass_flux) for nm in mdl.constituents])
        mdl.close_output(); mdl.engine.close()
        return out

    quiet = history()
    stop = threading.Event()
    errors = []

    def load():
        try:
            n = other['nreal'] + 1
            eng = make_engine(other, other_in)
            while not stop.is_set():
                eng.set_state(other_in[0, :n, :])
                for t in range(3):
                    eng.step(t, tol=1e-12)
                eng.get_state()
            eng.close()
        except Exception as exc:                      # pragma: no cover
            errors.append(exc)

    th = threading.Thread(target=load)
    th.start()
    try:
        busy = [history() for _ in range(3)]
    finally:
        stop.set(); th.join()
    assert not errors, errors
    for run in busy:
        for a, b in zip(run[0] + run[1], quiet[0] + quiet[1]):
            assert np.array_equal(a, b, equal_nan=True)
