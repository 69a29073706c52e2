#!/usr/bin/env python3
"""Per-step budget of a transport run from a rocprofv3 kernel trace (task 1a).

    python3 tools/trace_budget.py <..._kernel_trace.csv> [--steps N] [--label TEXT]

Reads the kernel trace (Kernel_Name, Start_Timestamp, End_Timestamp in ns), takes the LAST `--steps` steps (a step starts at its
k_begin_step launch -- one per cwr_step; older builds: k_prep_step / k_rhs) and prints, per step on average:
  * every kernel: launches per step, average duration, total per step;
  * the passes (k_sq_tiled) in particular: count x average = total;
  * idle time of the stream: gaps between consecutive kernels INSIDE a step (end of one -> start of the next), split into the gap in front
    of the step's first kernel after the previous step's last (the host round trip of the convergence check + the next step's
    launch) and all the others (launch gaps inside graph replays / between launches);
  * the wall time per step = first kernel of a step -> first kernel of the next.
"""
from __future__ import annotations

import argparse
import collections
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r'^void\s+', '', name)
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('cwr::', '').replace('(anonymous namespace)::', '')
    return name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--label', default='')
    args = ap.parse_args()
    rows = []
    with open(args.trace) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    rows.sort()
    marker = next((m for m in ('k_begin_step', 'k_prep_step', 'k_rhs') if any(n.startswith(m) for _, _, n in rows)), 'k_rhs')
    starts = [i for i, (_, _, n) in enumerate(rows) if n.startswith(marker)]
    if len(starts) < args.steps + 1:
        sys.exit(f'only {len(starts)} steps in the trace, need {args.steps + 1}')
    first = starts[-(args.steps + 1)]
    last = starts[-1]                      # the last step is cut off (its end is the end of the trace): use the N before it
    sel = rows[first:last]
    bounds = [s - first for s in starts[-(args.steps + 1):]]
    wall = (rows[last][0] - rows[first][0]) / args.steps
    per = collections.OrderedDict()
    for _, (s, e, n) in enumerate(sel):
        d = per.setdefault(n, [0, 0])
        d[0] += 1; d[1] += e - s
    busy = sum(v[1] for v in per.values()) / args.steps
    gap_head, gap_in, n_in = 0, 0, 0
    bset = set(bounds)
    for i in range(1, len(sel) + 1):
        cur_start = rows[first + i][0] if first + i < len(rows) else None
        if cur_start is None:
            break
        g = max(0, cur_start - sel[i - 1][1])
        if i in bset:
            gap_head += g
        else:
            gap_in += g; n_in += 1
    spans = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        spans.append(max(e for _, e, _ in sel[a:b]) - sel[a][0])
    span = sum(spans) / len(spans)
    print(f'== {args.label or args.trace}')
    print(f'   steps analysed: {args.steps}; step span (first kernel start -> last kernel end) {span / 1e3:9.1f} us = kernels busy {busy / 1e3:9.1f} us + idle between '
          f'kernels of a step {gap_in / args.steps / 1e3:7.1f} us ({n_in / args.steps:.1f} gaps of {gap_in / max(n_in, 1) / 1e3:.2f} us); idle between steps (check round '
          f'trip + next launch; + the state injection of a stand-alone rank) {gap_head / args.steps / 1e3:7.1f} us; first kernel to first kernel {wall / 1e3:9.1f} us')
    print(f'   {"kernel":60s} {"per step":>9s} {"avg us":>9s} {"us/step":>9s} {"share":>7s}')
    for n, (cnt, tot) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f'   {n:60s} {cnt / args.steps:9.2f} {tot / cnt / 1e3:9.2f} {tot / args.steps / 1e3:9.1f} {tot / args.steps / wall * 100:6.1f}%')


if __name__ == '__main__':
    main()
