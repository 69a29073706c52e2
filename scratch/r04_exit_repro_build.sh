#!/bin/bash
# Rebuilds the library the seven faulting runs of gpurun_out/r03w_small_gs.txt used: the tree of b50d8c8 (14:33, round 3) with the
# dropped experiment scratch/r03_small_gs.patch applied, -O1 -g, under scratch/_r03w_repro/ (not committed; delete after use).
set -e
cd "$(dirname "$0")/.."
rm -rf scratch/_r03w_repro && mkdir -p scratch/_r03w_repro
git archive b50d8c8 clearwater-riverine_amd clearwater_riverine_amd include oracle tests/models | tar -x -C scratch/_r03w_repro
(cd scratch/_r03w_repro && patch -p1 < ../r03_small_gs.patch)
grep -n "h_nb.assign" scratch/_r03w_repro/clearwater-riverine_amd/csrc/cwr_engine.hip || echo "(the patch removed eng->h_nb.assign / eng->h_edge.assign from cwr_create)"
/opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC -shared -w \
  scratch/_r03w_repro/clearwater-riverine_amd/csrc/cwr_engine.hip -o scratch/_r03w_repro/clearwater-riverine_amd/libcwr_transport.so
cp scratch/r04_exit_repro.py scratch/_r03w_repro/repro.py
