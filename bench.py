#!/usr/bin/env python3
"""Headline benchmark of the transport path: Mcell-updates/s on the synthetic 1 M-cell mesh.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
            --master-port P bench.py --gpus N --steps K --warmup W)

A "step" is one ClearwaterRiverine.update()-equivalent pass (transport.py:201-276) over all cells and
constituents: operator set-up, right-hand side, implicit solve, write-back, per-face mass flux.
Workload: BASELINE.json's 1 M-cell synthetic UNSTRUCTURED floodplain mesh (synthetic.bench_mesh: 1026 x 1026 jittered
quads of which 5 % are merged into 6-sided cells = exactly 1 000 000 real cells, 2 054 728 faces, 4 104 ghost cells,
locally shuffled cell and face numbering, CFL ~ 2.5, D = 0.5) with 16 DISTINCT constituents (own initial field,
fronts and boundary series each: synthetic.distinct_input_array) -- the north-star roofline target case;
--constituents 1 gives configs[3] literally, --mesh quad the structured 1000 x 1000 grid of round 1.
The whole flow field, boundary values and state are resident in HBM before the timed region; nothing is copied to
the host inside it.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
HBM_MEASURED_GBS = 6290.0      # measured float4 copy, same guide


def cpu_baseline(mesh: dict, input_col: np.ndarray, budget_s: float, max_steps: int):
    """The oracle (numpy COO assembly + scipy spsolve per constituent = the reference algorithm without its xarray
    overhead) timed on this host, single process, on the SAME mesh as the GPU run with one constituent.
    Imports and allocator are warmed on a tiny mesh; then steps of the full mesh are timed one by one until
    `max_steps` are done or `budget_s` seconds are used (at least one step)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import copy
    import cwr_oracle as oracle
    from clearwater_riverine_amd import synthetic
    tiny = synthetic.make_mesh(20, 10, 2, seed=1)
    oracle.derive_coefficients(tiny)
    oracle.OracleModel(tiny, {'c': synthetic.boundary_input_array(tiny, 1)[:, :, 0]}).update()
    m = copy.copy(mesh)
    oracle.derive_coefficients(m)
    model = oracle.OracleModel(m, {'c': np.ascontiguousarray(input_col)})
    n = m['nreal'] + 1
    times = []
    t_all = time.perf_counter()
    while len(times) < max_steps and (not times or time.perf_counter() - t_all + times[-1] <= budget_s):
        t0 = time.perf_counter()
        model.update()
        times.append(time.perf_counter() - t0)
    el = float(np.sum(times))
    return n * len(times) / el / 1e6, el, n, len(times)


def pmc_traffic(which: str, K: int, kernel: str = 'k_sq_tiled'):
    """HBM-side bytes per launch of the dominant kernel, measured NOW: two counter-only rocprofv3 passes (--pmc FETCH_SIZE,
    --pmc WRITE_SIZE; the guide's HBM section: separate passes, read = 2 x FETCH_SIZE x 1024 on gfx950, written =
    WRITE_SIZE x 1024) over tools/pmc_target.py -- two steps of this very workload -- as child processes.  Called before
    this process touches the GPU.  Returns (read, written) or None when rocprofv3 is not there or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None
    out = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='cwr_pmc_', dir='/tmp')
        try:
            env = dict(os.environ, TMPDIR='/tmp')
            subprocess.run(['rocprofv3', '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 'pmc', '--',
                            sys.executable, os.path.join(ROOT, 'tools', 'pmc_target.py'), which, str(K)],
                           check=True, capture_output=True, timeout=240, env=env, cwd='/tmp')
            tot, n = 0.0, 0
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kernel in r['Kernel_Name'] and r['Counter_Name'] == counter:
                        tot += float(r['Counter_Value']); n += 1
            if n == 0:
                return None
            out[counter] = tot / n
        except (subprocess.SubprocessError, OSError, KeyError, ValueError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(2 * out['FETCH_SIZE'] * 1024), int(out['WRITE_SIZE'] * 1024)


def _profiler_active() -> bool:
    """True when this process already runs under rocprofv3 (its tool library is preloaded): the counter passes below are
    themselves rocprofv3 children and must not be nested into a traced run."""
    pre = os.environ.get('LD_PRELOAD', '')
    return ('rocprofiler' in pre or 'rocprofv3' in pre or any(k.startswith(('ROCPROFILER_', 'ROCPROF_', 'ROCP_TOOL')) for k in os.environ))


def launch_ranks(n: int, argv: list[str], timeout_s: float = 1500.0) -> int:
    """`python bench.py --gpus N` without a launcher: THIS process (which has not imported torch or touched the GPU) starts
    N children of the same command line, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set --
    what `python -m torch.distributed.run --nproc-per-node N` would have set --, relays rank 0's JSON line, reaps every
    child and returns non-zero when any child failed (the others are then terminated: no orphan waits in a collective).
    The reference has no launcher to mirror (its K-loop is serial, transport.py:231-249)."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:                              # a free loopback port for the control plane
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    script = os.environ.get('CWR_BENCH_CHILD', os.path.abspath(__file__))     # (tests substitute a stub child)
    # HSA_ENABLE_IPC_MODE_LEGACY=0 (kept from the environment when set): this image's host driver supports dmabuf IPC only -- without it
    # RCCL's sharing of device buffers between the rank processes fails with `hipIpcGetMemHandle: invalid argument` (the image exports it
    # already; the launcher makes sure a rank started from a scrubbed environment has it too.  The torchrun route inherits the image's own.)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(n), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env, start_new_session=True,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    line, rc = None, 0
    deadline = time.monotonic() + timeout_s

    def stop_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except OSError:
                    pass
        t_end = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
                p.wait()

    # SIGTERM / SIGHUP to the launcher (an outer `timeout`, a scheduler): the ranks sit in sessions of their own and would outlive it,
    # holding their GPUs in a collective until their own alarm fires -- turn the signal into an exception so that the `finally` runs
    class _Stopped(Exception):
        pass

    def _on_signal(signum, _frame):
        raise _Stopped(signum)
    old_handlers = {sig: signal.signal(sig, _on_signal) for sig in (signal.SIGTERM, signal.SIGHUP)}
    try:
        import selectors
        sel = selectors.DefaultSelector()
        sel.register(procs[0].stdout, selectors.EVENT_READ)
        out_open = True
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0] if bad[0] > 0 else 1
                break
            if all(c == 0 for c in codes) and not out_open:
                break
            if time.monotonic() > deadline:
                print(f'bench.py: ranks still running after {timeout_s:.0f} s, stopping them', file=sys.stderr)
                rc = 124
                break
            if out_open:
                for key, _ in sel.select(timeout=0.2):
                    ln = key.fileobj.readline()
                    if ln == '':
                        sel.unregister(key.fileobj)
                        out_open = False
                    elif ln.lstrip().startswith('{'):
                        line = ln.rstrip('\n')
                    else:
                        sys.stderr.write(ln)
            else:
                time.sleep(0.05)
    except _Stopped as sig:
        print(f'bench.py: launcher got signal {sig.args[0]}, stopping the ranks', file=sys.stderr)
        rc = 128 + int(sig.args[0])
    finally:
        for sig_, h in old_handlers.items():
            signal.signal(sig_, signal.SIG_IGN)             # (a second signal must not interrupt the reaping)
        stop_all()
        for sig_, h in old_handlers.items():
            signal.signal(sig_, h)
    if rc == 0 and line is None:
        print('bench.py: rank 0 ended without a result line', file=sys.stderr)
        rc = 1
    if rc == 0:
        print(line, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--mesh', default='merged', choices=['merged', 'quad'],
                    help='merged: the unstructured bench mesh (5 %% of the quads merged into 6-sided cells, 10^6 cells); '
                         'quad: the structured 1000 x 1000 grid')
    ap.add_argument('--nx', type=int, default=0, help='custom grid (with --ny): nx x ny base quads, 5 %% merged unless --mesh quad')
    ap.add_argument('--ny', type=int, default=0)
    ap.add_argument('--inputs', default='distinct', choices=['distinct', 'scaled'],
                    help='distinct: every constituent has its own field / fronts / boundary series; scaled: multiples of one')
    ap.add_argument('--constituents', type=int, default=16)
    ap.add_argument('--tol', type=float, default=1e-12)
    ap.add_argument('--dt', type=float, default=40.0)
    ap.add_argument('--diffusion', type=float, default=0.5)
    ap.add_argument('--seed', type=int, default=4)
    ap.add_argument('--solver', default='auto', choices=['auto', 'jacobi', 'bicgstab'])
    ap.add_argument('--renumber', default='hilbert', choices=['hilbert', 'none'],
                    help='internal cell numbering (reference ids stay at the boundary)')
    ap.add_argument('--halo-depth', type=int, default=0,
                    help='N > 1: halo layers = Jacobi sweeps between two exchanges (0: from the per-rank size, distributed.auto_halo_depth)')
    ap.add_argument('--k-groups', type=int, default=1,
                    help='N > 1: the N GPUs as N / G contiguous cell ranges x G groups of constituents (distributed.GroupedTransport: a group is a '
                         'partitioned run of its constituents with a communicator of its own; groups never exchange).  1: N ranges of all constituents')
    ap.add_argument('--deterministic', action='store_true',
                    help='CWR_STEP_DETERMINISTIC: passes between two vectors, bitwise reproducible run to run (the default passes are chained in place)')
    ap.add_argument('--flow-window', type=int, default=0,
                    help='N = 1: keep only this many levels of the flow field on the device and upload one level per step beside the steps '
                         '(cwr_flow_window_open / _load; host arrays page-locked); 0: all levels resident (the headline configuration)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-rank-ceiling', action='store_true',
                    help='N > 1: skip the stand-alone step of every rank (the compute-side ceiling in the per-rank report)')
    ap.add_argument('--no-pmc', action='store_true', help='skip the two rocprofv3 counter passes that measure roofline.traffic')
    ap.add_argument('--cpu-budget-s', type=float, default=100.0, help='wall-clock budget of the CPU baseline leg')
    ap.add_argument('--cpu-steps', type=int, default=3)
    ap.add_argument('--windows', type=int, default=5,
                    help='the timed region (EXACTLY --steps steps between two barriers) is repeated this many times from the same '
                         'start state; value / ms_per_step are the MEDIAN window, every window is listed in the line')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (nothing below has run: no torch import, no GPU call)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        sys.exit(f'--gpus {args.gpus} but WORLD_SIZE is {world}: one process per GPU (torch.distributed.run --nproc-per-node '
                 f'{args.gpus}, or plain `python bench.py --gpus {args.gpus}`, which starts the ranks itself)')
    # roofline.traffic, measured in this run (children of a process that has NOT initialised the GPU yet)
    live_traffic = None
    if rank == 0 and world == 1 and not args.no_pmc and args.solver == 'auto' and not (args.nx or args.ny) and not _profiler_active():
        live_traffic = pmc_traffic(args.mesh, args.constituents)
    import torch                                       # device plumbing + control plane only
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU: the transport path has no CPU fallback')
    if 'CWR_BENCH_DEVICE' in os.environ:                # rehearsal of the N > 1 path on a one-GPU box (with CWR_RCCL_LIB)
        local_rank = int(os.environ['CWR_BENCH_DEVICE'])
    torch.cuda.set_device(local_rank)
    if world > 1:
        import signal
        signal.alarm(1500)                              # a lost rank must not leave the others waiting forever
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')  # one node: the control plane runs over loopback
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)

    from clearwater_riverine_amd import synthetic
    from clearwater_riverine_amd.distributed import GroupedTransport, PartitionedTransport, broadcast_bytes, group_layout
    from clearwater_riverine_amd.engine import TransportEngine

    K = args.constituents
    n_levels = args.warmup + args.steps + 1
    if args.nx and args.ny:
        mesh = synthetic.make_mesh(args.nx, args.ny, n_levels, seed=args.seed, dt=args.dt, diffusion_coefficient=args.diffusion,
                                   n_merge=0 if args.mesh == 'quad' else int(round(0.05 * args.nx * args.ny)))
        mesh_name = f'{args.nx}x{args.ny} ' + ('quad grid' if args.mesh == 'quad' else 'quads, 5 % merged into 6-sided cells')
        mesh_key = f'{args.nx}x{args.ny}' + ('' if args.mesh == 'quad' else '_merged')
    elif args.mesh == 'quad':
        mesh = synthetic.make_mesh(1000, 1000, n_levels, seed=args.seed, dt=args.dt, diffusion_coefficient=args.diffusion)
        mesh_name, mesh_key = 'structured 1000x1000 quad grid', '1000x1000'
    else:
        mesh = synthetic.bench_mesh(n_levels, dt=args.dt, diffusion_coefficient=args.diffusion)
        mesh_name = 'unstructured floodplain mesh: 1026x1026 jittered quads, 5 % merged into 6-sided cells, shuffled numbering'
        mesh_key = 'bench_merged_1m'
    # (distinct inputs: a provider of the slices a rank needs -- its own cells' initial rows, the series of the ghost cells it holds --
    # instead of the dense (T, ncell, K) block: 3.3 GB per rank at 1 M cells x 16, 13 GB at 4 M)
    inputs3 = (synthetic.DistinctInputs(mesh, K, seed=synthetic.BENCH_SEED) if args.inputs == 'distinct'
               else synthetic.boundary_input_array(mesh, K))
    n = mesh['nreal'] + 1

    uid = None
    G = max(1, args.k_groups)
    if world > 1 and G == 1:
        uid = broadcast_bytes(TransportEngine.comm_unique_id() if rank == 0 else None, 128, src=0)
    if G > 1:
        # R ranges x G constituent groups: one communicator per group, its id made by the group's range 0 and handed round over the control plane
        g_, r_, R_, k0_, k1_ = group_layout(rank, world, G, K)
        ids = [None] * world
        dist.all_gather_object(ids, TransportEngine.comm_unique_id() if (r_ == 0 and R_ > 1) else None)
        pt = GroupedTransport(mesh, inputs3, rank, world, G, device=local_rank, unique_id=ids[g_ * R_], halo_depth=args.halo_depth,
                              renumber=None if args.renumber == 'none' else args.renumber, flow_window=args.flow_window or None)
    else:
        pt = PartitionedTransport(mesh, inputs3, rank, world, device=local_rank, unique_id=uid, halo_depth=args.halo_depth,
                                  renumber=None if args.renumber == 'none' else args.renumber, flow_window=args.flow_window or None)
    eng = pt.engine
    K_loc = pt.K                                          # constituents THIS rank carries (K / G with constituent groups)
    world_r = world // G                                  # cell ranges = ranks of one communicator

    def barrier():
        eng.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    iters = []
    for t in range(args.warmup):
        pt.step(t, tol=args.tol, mass_flux=True, solver=args.solver, deterministic=args.deterministic)
    saved = eng.get_state()[: pt.local.n_core].copy()       # state at the start of the timed region
    # The timed region -- barrier, EXACTLY --steps steps, barrier -- is repeated --windows times from the same start state
    # (restored outside the timed region); every window takes the MAX over the ranks, the line reports the median window.
    windows = []
    for w in range(max(1, args.windows)):
        if w > 0:
            eng.set_state(saved)
        pt.fill_window(args.warmup)                       # (windowed flow field: a repeated window starts with its levels in the ring, as the
        iters = []                                        #  first one does behind the warm-up steps; barrier() sends and awaits the uploads)
        barrier()
        t0 = time.perf_counter()
        for t in range(args.warmup, args.warmup + args.steps):
            r = pt.step(t, tol=args.tol, mass_flux=True, solver=args.solver, deterministic=args.deterministic)
            iters.append({'sweeps': r.sweeps, 'bicgstab': r.iterations} if world == 1 else
                         {'sweeps': r.sweeps, 'bicgstab': r.iterations, 'exchanges': r.exchanges, 'overlapped': r.overlapped, 'checks': r.checks})
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        windows.append(el)
    max_resid = r.max_rel_residual
    elapsed = float(np.median(windows))

    # ---- roofline of the dominant kernel (the face-flux operator), HIP events on the engine's stream ----
    roofline = None
    eng.set_state(saved)                                 # every rank replays the timed steps, event-timed
    pt.fill_window(args.warmup)
    eng.profile_read()
    for t in range(args.warmup, args.warmup + args.steps):
        pt.step(t, tol=args.tol, mass_flux=True, profile=True, solver=args.solver, deterministic=args.deterministic)
    launches, total_us = eng.profile_read()
    # ---- N > 1: what every rank's step consisted of, so that a scaling run explains itself ----
    # computed / owned rows, passes x mean pass us, the exchanges (alone / beside compute) and all-reduces on the communication stream, the
    # host's wall time inside the checks -- from the event-timed replay above -- and the COMPUTE-SIDE CEILING: the same rank stepped alone
    # (tools/rank_step_profile.py's method: its real partition with a one-rank communicator, nothing exchanged) through the sweep count
    # of the real run.  single-GPU step / slowest rank's stand-alone step is what N GPUs could reach if communication were free.
    ranks_report = None
    if world > 1:
        cp = eng.comm_profile_read()
        _ok, ntiles, tgrid, trows = eng.tiling_info()
        lm = pt.local
        sw_med = int(np.median([i['sweeps'] for i in iters])) if iters else 0
        mine = {'rank': rank, 'cell_range': rank % world_r, 'constituent_group': rank // world_r, 'constituents': K_loc, 'owned_rows': int(lm.n_core), 'computed_rows': int(lm.n_rows), 'halo_rows': int(lm.n_halo), 'peers': int(len(lm.peers)),
                'halo_depth': int(lm.depth), 'tiles': int(ntiles), 'tile_rows': int(trows), 'grid': int(tgrid),
                'timed_region_ms_per_step': [round(1000.0 * w / args.steps, 3) for w in windows],
                'passes_per_step': round(launches / args.steps, 2), 'mean_pass_us': round(total_us / launches, 2) if launches else None,
                'sweeps_per_step_median': sw_med, 'steps_profiled': args.steps}
        mine.update({k: (round(v, 1) if isinstance(v, float) else v) for k, v in cp.items()})
        mine['standalone_ms_per_step'] = None
        if not args.no_rank_ceiling:
            dist.barrier()
            try:
                os.environ['CWR_TEST_FIXED_SWEEPS'] = str(max(3, sw_med | 1))
                from clearwater_riverine_amd.distributed import ConstituentSlice
                in_alone = inputs3 if G == 1 else (ConstituentSlice(inputs3, pt.k0, pt.k1) if hasattr(inputs3, 'ghost_columns') else inputs3[:, :, pt.k0:pt.k1])
                alone = PartitionedTransport(mesh, in_alone, rank % world_r, world_r, device=local_rank, halo_depth=args.halo_depth,
                                             renumber=None if args.renumber == 'none' else args.renumber, standalone=world_r > 1)
                os.environ.pop('CWR_TEST_FIXED_SWEEPS', None)
                for t in range(args.warmup):
                    alone.step(t, tol=args.tol, mass_flux=True, solver=args.solver, deterministic=args.deterministic)
                alone.engine.synchronize()
                t0 = time.perf_counter()
                for t in range(args.warmup, args.warmup + args.steps):
                    alone.step(t, tol=args.tol, mass_flux=True, solver=args.solver, deterministic=args.deterministic)
                alone.engine.synchronize()
                mine['standalone_ms_per_step'] = round(1000.0 * (time.perf_counter() - t0) / args.steps, 4)
                mine['standalone_sweeps'] = max(3, sw_med | 1)
                alone.engine.close()
            except Exception as exc:                     # (a report, not the measurement: say why and go on)
                os.environ.pop('CWR_TEST_FIXED_SWEEPS', None)
                mine['standalone_error'] = f'{type(exc).__name__}: {str(exc)[:160]}'
            dist.barrier()
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ranks_report = gathered
    if rank == 0:
        b_r, b_w = eng.apply_bytes()
        kernel_name = {4: 'k_apply<VW,4>: fused Jacobi sweep of the face-flux operator',
                       5: 'k_apply<VW,5>: J^2 pass (two Jacobi iterations per launch)',
                       6: 'k_sq_tiled<VW>: J^2 pass with the x tile staged in LDS, software-pipelined (two Jacobi iterations = two operator applies per launch)',
                       7: 'k_small_jacobi: one-launch LDS-resident solve'}.get(r.sweep_kernel, 'k_apply<VW,1>: BiCGSTAB product')
        pt.fill_window(args.warmup)                       # (windowed flow field: the level the timing loop uses back into the ring)
        back_to_back_us = eng.time_apply(args.warmup, reps=50) if world == 1 else None
        traffic = traffic_rw = None
        traffic_source = None
        if live_traffic is not None and r.sweep_kernel == 6:
            traffic_rw, traffic = live_traffic, sum(live_traffic)
            traffic_source = 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over tools/pmc_target.py in this run'
        try:                                             # otherwise the committed PMC measurement of this exact config, if any
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fh:
                ent = json.load(fh).get(mesh_key, {}).get(str(K)) if world == 1 and args.solver == 'auto' else None
            if ent and traffic is None:
                traffic_rw = (int(ent['read']), int(ent['written']))
                traffic = sum(traffic_rw)
                traffic_source = 'profiles/pmc_traffic.json (committed rocprofv3 --pmc measurement of this configuration)'
        except (OSError, KeyError, TypeError, ValueError):
            pass
        if launches > 0:
            avg_us = total_us / launches
            # ALGORITHMIC bytes of a launch = SURVEY.md section 8(d)'s per-apply figure x the operator applies one launch performs:
            #   B_r(apply) = 24 E + 8 K n + 12 n + 4 + 4 (2 E_int + E_ghost)      (faces, x once, diagonal, CSR adjacency)
            # a J^2 pass advances two Jacobi iterations = two applies (its tile-local re-applications are not counted).
            lm = pt.local
            f2 = np.asarray(lm.face2)
            E_loc = len(f2)
            E_ghost = int(np.count_nonzero(f2 >= lm.n_rows + lm.n_halo))
            n_loc = lm.n_rows
            survey_apply = 24 * E_loc + 8 * K_loc * n_loc + 12 * n_loc + 4 + 4 * (2 * (E_loc - E_ghost) + E_ghost)
            applies = 2 if r.sweep_kernel in (5, 6) else 1
            alg = applies * survey_apply
            achieved = alg / (avg_us * 1e-6) / 1e9
            kernel_rate = b_r / (avg_us * 1e-6) / 1e9
            roofline = {
                'bound': 'hbm', 'kernel': kernel_name,
                'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                'traffic_read': None if traffic_rw is None else traffic_rw[0],
                'traffic_written': None if traffic_rw is None else traffic_rw[1], 'traffic_source': traffic_source,
                'algorithmic_bytes': alg, 'applies_per_launch': applies, 'survey_bytes_per_apply': survey_apply,
                'avg_launch_us': round(avg_us, 2), 'launches_timed': launches,
                # the bytes THIS kernel has to read / write per launch (pre-multiplied J^2 entries: fewer than two applies' worth)
                'kernel_bytes_read': b_r, 'kernel_bytes_written': b_w,
                'achieved_kernel_bytes_read': round(kernel_rate, 1), 'frac_kernel_bytes_read': round(kernel_rate / HBM_PEAK_GBS, 4),
                'achieved_read_plus_write': round((b_r + b_w) / (avg_us * 1e-6) / 1e9, 1),
                'frac_rw_of_measured_stream_peak': round((b_r + b_w) / (avg_us * 1e-6) / 1e9 / HBM_MEASURED_GBS, 4),
                'back_to_back_launch_us': None if back_to_back_us is None else round(back_to_back_us, 2),
            }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        v, el, nn, done = cpu_baseline(mesh, inputs3.column(0) if hasattr(inputs3, 'column') else inputs3[:, :, 0], args.cpu_budget_s, args.cpu_steps)
        cpu = {'value': round(v, 5), 'unit': 'Mcell-updates/s', 'cores': 1, 'kind': 'port',
               'sample': f'the same mesh ({nn} cells, {mesh_name}), constituent 0 only, {done} step(s) of {el / done:.1f} s each '
                         f'(budget {args.cpu_budget_s:.0f} s; imports warmed on a 200-cell mesh); numpy COO assembly + '
                         f'scipy.sparse.linalg.spsolve per constituent (single-threaded SuperLU); host has {os.cpu_count()} cores',
               'host_cores': os.cpu_count()}

    if rank == 0:
        chained = r.chained == 1
        walked = r.chained == 2
        value = n * K * args.steps / elapsed / 1e6
        line = {
            'metric': 'Mcell-updates/s', 'value': round(value, 2), 'unit': 'Mcell-updates/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1000.0 * elapsed / args.steps, 3), 'higher_is_better': True,
            'windows': {'n': len(windows), 'statistic': 'median', 'ms_per_step': [round(1000.0 * w / args.steps, 3) for w in windows],
                        'value_min': round(n * K * args.steps / max(windows) / 1e6, 2),
                        'value_max': round(n * K * args.steps / min(windows) / 1e6, 2)},
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'synthetic {mesh_name} ({n} cells, {len(mesh["edges_face1"])} faces), {K} '
                                   f'{"distinct " if args.inputs == "distinct" and K > 1 else ""}constituents, implicit upwind '
                                   f'advection-diffusion step, dt={args.dt} s, D={args.diffusion}',
                       'cells': n, 'faces': int(len(mesh['edges_face1'])), 'constituents': K,
                       'numbering': pt.numbering + (' + tile-balanced windows' if pt.numbering != 'reference' else ''),
                       'partition': f'contiguous cell ranges x{world_r}' + (f' x {G} groups of {K_loc} constituents (no exchange between groups)' if G > 1 else '') + (f', halo depth {pt.local.depth}' if world_r > 1 else ''),
                       'tol': args.tol, 'flow_field': (f'ring of {args.flow_window} levels, one level uploaded per step beside the steps'
                                                       if args.flow_window and world == 1 else 'all levels resident in HBM')},
            'solver': {'method': ('J^2 passes of fused Jacobi sweeps, tiles chained along the flow and relaxed in place (block Gauss-Seidel '
                                  'along the flow, no inter-block waiting)' if chained else
                                  'J^2 passes of fused Jacobi sweeps between two vectors, tiles chained along the flow (a tile takes its '
                                  'predecessor\'s rows from LDS): deterministic' if walked else
                                  'block-asynchronous (ping-pong) J^2 passes of fused Jacobi sweeps') + '; exact closing sweep; BiCGSTAB '
                                 'fallback; K systems batched',
                       'chained_passes': chained, 'deterministic_chained_passes': walked, 'tile_local_applications': r.local_reps, 'iterations_per_step': iters,
                       'max_rel_residual': max_resid},
            'roofline': roofline, 'cpu_baseline': cpu,
        }
        if ranks_report is not None:
            line['ranks'] = ranks_report
            alone_ms = [r.get('standalone_ms_per_step') for r in ranks_report]
            if all(a for a in alone_ms):
                line['compute_side_ceiling'] = {'slowest_rank_standalone_ms_per_step': max(alone_ms), 'measured_ms_per_step': round(1000.0 * elapsed / args.steps, 3),
                                                'note': 'a rank stepped alone through the real run\'s sweep count (one-rank communicator, nothing exchanged): measured - standalone = what communication and waiting cost'}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
