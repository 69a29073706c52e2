"""A process that ends with live engines / facades must end cleanly.

pytest, smoke() and bench.py all release their engines before they exit, so none of them says anything about a host program that
simply ends -- with a TransportEngine still referenced from a module global, a facade whose history blocks are page-locked, or a
streamed-output writer thread still attached.  Every mode of tools/exit_probe.py runs as a FRESH child process here; the child
must exit with status 0 and an empty stderr (no traceback from a finalizer, no 'Exception ignored in', no fault)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, 'tools', 'exit_probe.py')


@pytest.mark.parametrize('mode', ['engine', 'engine_global', 'facade', 'facade_closed', 'facade_stream'])
def test_process_exits_cleanly_with_live_objects(gpu_lib, mode):
    env = dict(os.environ)
    env.pop('CWR_VERBOSE', None)
    res = subprocess.run([sys.executable, PROBE, mode], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, (res.returncode, res.stdout[-2000:], res.stderr[-2000:])
    assert res.stdout.strip().endswith(f'{mode} done'), res.stdout[-2000:]
    assert res.stderr.strip() == '', res.stderr[-2000:]


def test_two_facades_in_a_row_release_their_page_locks_before_their_pages(gpu_lib):
    """The facade's history blocks are page-locked (cwr_host_register) anonymous mappings.  When a facade is replaced by the next
    one in the same process its locks must be released BEFORE its pages are unmapped: the next facade's blocks may land on the
    same addresses, and a stale registration of a recycled range makes the copies of update() fail or land in the wrong pages
    (tests/models/ohio_like.py builds five facades in a row this way)."""
    code = (
        "import sys, os; sys.path.insert(0, %r)\n"
        "import numpy as np, clearwater_riverine_amd as cw\n"
        "last = None\n"
        "for rep in range(4):\n"
        "    mesh = cw.synthetic.make_mesh(40, 16, 5, seed=3)\n"
        "    inputs3 = cw.synthetic.boundary_input_array(mesh, 2)\n"
        "    model = cw.ClearwaterRiverine(mesh=dict(mesh), input_arrays={f'c{k}': inputs3[:, :, k].copy() for k in range(2)})\n"
        "    for _ in range(3): model.update()\n"
        "    got = np.array(model.mesh['c0'][3])\n"
        "    assert last is None or np.array_equal(got, last, equal_nan=True), rep\n"
        "    last = got\n"
        "print('ok')\n" % ROOT)
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0 and res.stdout.strip() == 'ok' and res.stderr.strip() == '', (res.returncode, res.stdout[-1000:], res.stderr[-2000:])
