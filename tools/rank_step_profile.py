#!/usr/bin/env python3
"""One rank of a partitioned run, stepped ALONE: the compute side of that rank's step (task 1).

    python3 tools/rank_step_profile.py --mesh 1m --K 16 --rank 3 --world 8 [--steps 10 --warmup 4]
    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -o NAME -- python3 tools/rank_step_profile.py ...
    python3 tools/trace_budget.py DIR/**/NAME_kernel_trace.csv --steps 10

Builds rank r of `world` of the REAL partition of the synthetic 1 M-cell (--mesh 1m) or 4 M-cell (--mesh 4m) mesh -- the same
numbering rule, halo depth and alignment as bench.py --gpus N: distributed.PartitionedTransport(standalone=True) -- core +
replayed layers + read-only layer, and steps that local engine with NO communicator: the read-only halo rows are frozen at the
initial field (written once through the exported state pointer), nothing is exchanged or all-reduced.  --world 1: the single-GPU
step of the same mesh, for the table's reference line.  Prints one line: sizes, tiling, ms per step, sweeps per step.
"""
from __future__ import annotations

import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mesh', default='1m', help='1m | 4m | sqN (an N x N jittered, 5 %% merged mesh)')
    ap.add_argument('--K', type=int, default=16)
    ap.add_argument('--rank', type=int, default=0)
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--depth', type=int, default=0, help='halo depth (0: distributed.auto_halo_depth)')
    ap.add_argument('--dt', type=float, default=40.0)
    ap.add_argument('--no-flux', action='store_true')
    ap.add_argument('--deterministic', action='store_true')
    args = ap.parse_args()

    import clearwater_riverine_amd as cw
    from clearwater_riverine_amd.distributed import PartitionedTransport
    T = args.warmup + args.steps + 1
    if args.mesh in ('1m', '4m'):
        mesh = cw.synthetic.bench_mesh(T, dt=args.dt, scale=1 if args.mesh == '1m' else 2)
    else:
        nx = int(args.mesh[2:])
        mesh = cw.synthetic.make_mesh(nx, nx, T, seed=4, dt=args.dt, diffusion_coefficient=0.5, n_merge=int(0.05 * nx * nx))
    inputs3 = cw.synthetic.distinct_input_array(mesh, args.K, seed=cw.synthetic.BENCH_SEED)
    pt = PartitionedTransport(mesh, inputs3, args.rank, args.world, halo_depth=args.depth, standalone=True)
    eng, lm = pt.engine, pt.local
    if pt.standalone:
        # every real row outside the core starts from the initial field too (an exchange would have delivered it): the replayed
        # layers and the read-only layer, written once through the exported state pointer
        n = mesh['nreal'] + 1
        ref = lm.cell_global if pt.order is None else np.where(lm.cell_global < n, pt.order[np.minimum(lm.cell_global, n - 1)], lm.cell_global)
        halo = np.ascontiguousarray(inputs3[0, ref[lm.n_core:lm.n_rows + lm.n_halo], :], dtype=np.float64)
        ptr, _stream = eng.state_device_ptr()
        hip = ctypes.CDLL('libamdhip64.so')
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        eng.synchronize()
        rc = hip.hipMemcpy(ctypes.c_void_p(ptr + lm.n_core * args.K * 8), halo.ctypes.data_as(ctypes.c_void_p), halo.nbytes, 1)
        assert rc == 0, f'hipMemcpy of the halo rows failed ({rc})'
    kw = dict(tol=1e-12, mass_flux=not args.no_flux, deterministic=args.deterministic)
    for t in range(args.warmup):
        pt.step(t, **kw)
    eng.synchronize()
    t0 = time.perf_counter()
    sw, chk = [], []
    for t in range(args.warmup, args.warmup + args.steps):
        r = pt.step(t, **kw)
        sw.append(r.sweeps); chk.append(r.checks)
    eng.synchronize()
    el = (time.perf_counter() - t0) / args.steps
    ok, ntiles, grid, TR = eng.tiling_info()
    print(f'RANKSTEP mesh={args.mesh} K={args.K} rank={args.rank}/{args.world} depth={lm.depth} numbering={pt.numbering} core={lm.n_core} '
          f'computed={lm.n_rows} halo={lm.n_halo} peers={len(lm.peers)} tiles={ntiles}x{TR} grid={grid} ({ntiles / max(grid, 1):.2f}/block) '
          f'kernel={r.sweep_kernel} chained={r.chained} reps={r.local_reps} ms_per_step={el * 1e3:.4f} sweeps={min(sw)}-{max(sw)} '
          f'checks={min(chk)}-{max(chk)} flags={r.flags}', flush=True)
    eng.close()


if __name__ == '__main__':
    main()
