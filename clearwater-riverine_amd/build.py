"""Build the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'cwr_engine.hip')
# (round 6: cwr_engine.hip is ONE translation unit cut along its section banners into csrc/cwr_engine_*.hpp: every header under csrc/ is a dependency)
DEPS = [SRC, os.path.join(os.path.dirname(HERE), "include", "cwr_transport.h")] + \
       sorted(os.path.join(HERE, 'csrc', f) for f in os.listdir(os.path.join(HERE, 'csrc')) if f.endswith('.hpp'))
OUT = os.path.join(HERE, 'libcwr_transport.so')

FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-munsafe-fp-atomics', '-fPIC', '-shared',
         '-Wall', '-Wno-unused-function', '-Wno-unused-value', '-Wno-unused-result']


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + [SRC, '-o', OUT]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
