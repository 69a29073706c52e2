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


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    m = np.isfinite(b)
    assert np.array_equal(np.isnan(a), np.isnan(b)), 'NaN pattern differs'
    if not m.any():
        return 0.0
    return float(np.max(np.abs(a[m] - b[m])) / max(np.max(np.abs(b[m])), 1e-300))
