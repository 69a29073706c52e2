#!/usr/bin/env python3
"""One rank of a partitioned run, stepped ALONE: the compute side of that rank's step (task 1).

    python3 tools/rank_step_profile.py --mesh 1m --K 16 --rank 3 --world 8 [--steps 10 --warmup 4]
    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -o NAME -- python3 tools/rank_step_profile.py ...
    python3 tools/trace_budget.py DIR/**/NAME_kernel_trace.csv --steps 10

Builds rank r of `world` of the REAL partition of the synthetic 1 M-cell (--mesh 1m) or 4 M-cell (--mesh 4m) mesh -- the same
numbering rule, halo depth and alignment as bench.py --gpus N: distributed.PartitionedTransport(standalone=True) -- core +
replayed layers + read-only layer, and steps that local engine with a communicator of ONE rank and no peers (cwr_attach_comm's
stand-alone form: the launch structure of a rank, nothing exchanged or all-reduced).  Its halo data is PERFECT: the single-GPU
solution of the same mesh, computed first in this process, is written into the rank's rows before every step (level t into the rows
it computes, level t + 1 into its read-only layer), outside the timed calls.  --world 1: the single-GPU step of the same mesh, for
the table's reference line.  Prints one line: sizes, tiling, ms per step (each step timed on its own), sweeps per step, and the
largest difference of the rank's core rows from the single-GPU solution.
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
    ap.add_argument('--fixed-sweeps', type=int, default=0,
                    help='stand-alone ranks: run exactly this many sweeps per step (one batch) and take the result as it is '
                         '(CWR_TEST_FIXED_SWEEPS): the launch sequence of a converging step of that length.  For ranks whose own '
                         'iteration cannot converge alone (nobody refreshes the outermost computed layer of the partner vector, which '
                         'an upstream halo feeds straight into the core)')
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
    kw = dict(tol=1e-12, mass_flux=not args.no_flux, deterministic=args.deterministic)
    n = mesh['nreal'] + 1
    truth = None
    if args.fixed_sweeps and args.world > 1:
        os.environ['CWR_TEST_FIXED_SWEEPS'] = str(args.fixed_sweeps)
    pt = PartitionedTransport(mesh, inputs3, args.rank, args.world, halo_depth=args.depth, standalone=True)
    os.environ.pop('CWR_TEST_FIXED_SWEEPS', None)           # (the reference engine below converges for real)
    eng, lm = pt.engine, pt.local
    if args.world > 1:
        # what the other ranks would deliver: the single-GPU solution of every level (a second engine over the whole mesh, gone before the
        # rank steps; built AFTER the rank's communicator: RCCL's initialisation fails in a process that has already run an engine)
        ref = PartitionedTransport(mesh, inputs3, 0, 1)
        truth = [np.ascontiguousarray(inputs3[0, :n, :], dtype=np.float64)]
        for t in range(args.warmup + args.steps):
            ref.step(t, **kw)
            truth.append(ref.gather_state())
        ref.engine.close()
    hip = rows_ref = None
    if pt.standalone:
        # A rank stepped alone is the rank's own iteration with PERFECT halo data: before step t every row it computes holds the
        # solution of level t (what the start-of-step exchange delivers) and its read-only layer the solution of level t + 1 (what the
        # exchanges of the step converge to) -- written through the exported state pointer, outside the timed calls.  (Freezing the
        # read-only layer at old values instead makes the local problem inconsistent with the step: the sweeps stall.)
        assert eng.state_row_stride() == args.K
        rows_ref = lm.cell_global[:lm.n_real] if pt.order is None else pt.order[lm.cell_global[:lm.n_real]]
        ptr, _stream = eng.state_device_ptr()
        hip = ctypes.CDLL('libamdhip64.so')
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

    def inject(t):
        if hip is None:
            return
        buf = np.empty((lm.n_real, args.K), np.float64)
        buf[:lm.n_rows] = truth[t][rows_ref[:lm.n_rows]]
        buf[lm.n_rows:] = truth[t + 1][rows_ref[lm.n_rows:]]
        eng.synchronize()
        rc = hip.hipMemcpy(ctypes.c_void_p(ptr), buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes, 1)
        assert rc == 0, f'hipMemcpy of the rank rows failed ({rc})'

    for t in range(args.warmup):
        inject(t)
        pt.step(t, **kw)
    sw, chk, el = [], [], 0.0
    eng.synchronize()
    t_all = time.perf_counter()
    for t in range(args.warmup, args.warmup + args.steps):
        if hip is not None:                              # (a stand-alone rank: every step timed on its own, around the injection)
            inject(t)
            eng.synchronize()
            t0 = time.perf_counter()
        r = pt.step(t, **kw)
        if hip is not None:
            eng.synchronize()
            el += time.perf_counter() - t0
        sw.append(r.sweeps); chk.append(r.checks)
    eng.synchronize()
    el = (el if hip is not None else time.perf_counter() - t_all) / args.steps      # (world 1: the steps back to back, as bench.py times them)
    err = 0.0
    if truth is not None:
        got = eng.get_state()[:lm.n_core]
        want = truth[args.warmup + args.steps][rows_ref[:lm.n_core]]
        err = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
    ok, ntiles, grid, TR = eng.tiling_info()
    print(f'RANKSTEP mesh={args.mesh} K={args.K} rank={args.rank}/{args.world} depth={lm.depth} numbering={pt.numbering} core={lm.n_core} '
          f'computed={lm.n_rows} halo={lm.n_halo} peers={len(lm.peers)} tiles={ntiles}x{TR} grid={grid} ({ntiles / max(grid, 1):.2f}/block) '
          f'kernel={r.sweep_kernel} chained={r.chained} reps={r.local_reps} ms_per_step={el * 1e3:.4f} sweeps={min(sw)}-{max(sw)} '
          f'checks={min(chk)}-{max(chk)} flags={r.flags} core_vs_single_gpu={err:.1e} fixed_sweeps={args.fixed_sweeps}', flush=True)
    eng.close()


if __name__ == '__main__':
    main()
