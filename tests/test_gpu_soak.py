"""A seeded soak of the whole step against the oracle (tests/soak_step.py) as one GPU test: 40 random small meshes -- sizes, 6- / 8-sided and
dry cells, CFL 0.6 ... 200, K in {1 ... 16}, grid caps that make small meshes chain, numberings, deterministic / in-place passes, bent
channels -- each through the C ABI, max-norm <= 1e-9 and the element-wise bar of util.rel_err.  A fresh interpreter: the cases set and
clear environment knobs that engines read at creation."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.mid_mesh_default
def test_forty_random_steps_match_the_oracle(gpu_lib):
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, 'soak_step.py'), '40', '2026'], capture_output=True, text=True, timeout=900)
    tail = '\n'.join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0, tail
    assert '40 of 40 cases ok' in r.stdout, tail
    # the soak reaches the code it was written beside: mid-size meshes through the several-workgroups solver (kernel 7), others of the
    # same sizes through the tiled passes (CWR_NO_SMALL drawn)
    import re
    mid = [(int(m.group(1)), int(m.group(2))) for m in re.finditer(r': n=(\d+) kernel (\d+) ', r.stdout)]
    mid = [kk for n, kk in mid if 4096 < n <= 24576]
    assert mid.count(7) >= 3 and mid.count(6) >= 3, r.stdout
