"""A seeded soak of the whole step against the oracle (tests/soak_step.py) as one GPU test: 40 random small meshes -- sizes, 6- / 8-sided and
dry cells, CFL 0.6 ... 200, K in {1 ... 16}, grid caps that make small meshes chain, numberings, deterministic / in-place passes, bent
channels -- each through the C ABI, max-norm <= 1e-9 and the element-wise bar of util.rel_err.  A fresh interpreter: the cases set and
clear environment knobs that engines read at creation."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_forty_random_steps_match_the_oracle(gpu_lib):
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, 'soak_step.py'), '40', '2026'], capture_output=True, text=True, timeout=900)
    tail = '\n'.join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0, tail
    assert '40 of 40 cases ok' in r.stdout, tail
