import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'mid_mesh_default: meshes of 4 097 .. 24 576 cells take what the engine takes by default -- the one-launch '
                                       'solver with several workgroups per constituent (round 5) -- instead of the tiled passes')


@pytest.fixture(autouse=True)
def tiled_passes_for_mid_size_meshes(request, monkeypatch):
    """Since round 5 single-GPU engines of up to 24 576 cells take the one-launch solver (k_small_jacobi, several parts).  The many
    tests written on meshes of 5-15 k cells to exercise the TILED passes (which stay the product path above that size and on
    every rank of a partition) keep doing so: CWR_SMALL_MAX_CELLS=0 unless a test is marked mid_mesh_default.  Meshes of up to
    4 096 cells are not affected."""
    if request.node.get_closest_marker('mid_mesh_default') is None:
        monkeypatch.setenv('CWR_SMALL_MAX_CELLS', '0')


@pytest.fixture(scope='session')
def gpu_lib():
    """The built HIP extension, loaded; GPU tests fail loudly if it is missing (no CPU fallback)."""
    import clearwater_riverine_amd as cw
    return cw.load_library()
