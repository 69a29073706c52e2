"""Shared helpers of the test-suite: fixtures -> oracle mesh dicts, input arrays, engines."""
import copy
import os

import numpy as np

import cwr_oracle as oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_plan(plan: str, D: float):
    """Oracle mesh dict of a reference HDF fixture + the (T, ncell) input array of the reference's
    own test set-up: IC from the CSV, BC = CSV value on the ghost cell of every BC-line face."""
    z = np.load(os.path.join(GOLDEN, f'{plan}_inputs.npz'))
    mesh = oracle.mesh_from_fixture(z, D)
    T = len(mesh['time_seconds'])
    ghosts = mesh['edges_face2'][z['bc_face_index']]
    bc = {int(g): np.full(T, 100.0) for g in ghosts}
    inp = oracle.build_input_array(mesh, z['ic_cell_index'], z['ic_concentration'], bc)
    return mesh, inp, z


def multi_inputs(inp: np.ndarray, K: int, seed: int = 0) -> np.ndarray:
    """(T, ncell) -> (T, ncell, K): constituent k is the base array scaled by a fixed positive factor."""
    rng = np.random.default_rng(seed)
    scale = np.concatenate([[1.0], 0.25 + 2.0 * rng.random(K - 1)]) if K > 1 else np.ones(1)
    return inp[:, :, None] * scale[None, None, :]


def oracle_run(mesh, inputs3, n_steps, overrides=None):
    """Run the oracle for n_steps with K constituents; returns the OracleModel."""
    K = inputs3.shape[2]
    model = oracle.OracleModel(copy.copy(mesh), {f'c{k}': inputs3[:, :, k].copy() for k in range(K)})
    for s in range(n_steps):
        model.update(overrides.get(s) if overrides else None)
    return model


def rel_err(a, b, ew_rtol=1e-6, ew_atol=1e-12):
    """Parity metric of the concentration tests.  Returns the max-norm relative error max|a-b| / max|b| (which the
    callers hold to 1e-9) AND asserts the north-star bar element by element:
        |a_i - b_i| <= ew_rtol |b_i| + ew_atol max|b|        (1e-6 relative, absolute floor 1e-12 of the peak)
    so that a plume-front cell many decades below the peak cannot be wrong by its own size and pass.
    The NaN pattern (ghost cells without a boundary value) must be identical."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    m = np.isfinite(b)
    assert np.array_equal(np.isnan(a), np.isnan(b)), 'NaN pattern differs'
    if not m.any():
        return 0.0
    peak = max(np.max(np.abs(b[m])), 1e-300)
    err = np.abs(a[m] - b[m])
    bar = ew_rtol * np.abs(b[m]) + ew_atol * peak
    if np.any(err > bar):
        i = int(np.argmax(err / bar))
        raise AssertionError(f'element-wise bar violated at {int(np.count_nonzero(err > bar))} of {err.size} entries: worst '
                             f'|a-b| = {err[i]:.3e} where |b| = {abs(b[m][i]):.3e} (peak {peak:.3e}), {err[i] / bar[i]:.2f} x the bar')
    return float(np.max(err) / peak)


def flux_err(a, b):
    """Mass-flux arrays: a diffusive flux d (c_N - c_P) dt is a difference of nearly equal concentrations, so its own
    relative error is unbounded by any bar on c; the element-wise absolute floor is 1e-9 of the largest flux."""
    return rel_err(a, b, ew_rtol=1e-6, ew_atol=1e-9)
