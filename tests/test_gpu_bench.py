"""GPU test of bench.py's contract: one JSON line with the metric, the timing fields, `roofline` and `cpu_baseline`; and the
self-launched N > 1 path (two ranks on the one GPU through the RCCL stand-in).  Small meshes: seconds, not minutes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _line(cmd, env=None, timeout=600):
    p = subprocess.run([sys.executable, BENCH] + cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_line_carries_the_contract_fields(gpu_lib):
    d = _line(['--nx', '300', '--ny', '300', '--steps', '3', '--warmup', '2', '--windows', '3', '--no-pmc', '--cpu-steps', '1',
               '--cpu-budget-s', '20'])
    assert d['metric'] == 'Mcell-updates/s' and d['unit'] == 'Mcell-updates/s' and d['higher_is_better'] is True
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 2 and d['dtype'] == 'f64' and d['data'] == 'synthetic'
    assert d['vs_baseline'] is None and d['scaling'] in ('strong', 'weak') and 'workload' in d['config']
    n, K = d['config']['cells'], d['config']['constituents']
    # value, ms_per_step and the window list are consistent with ONE window of exactly `steps` steps (the median one)
    assert d['windows']['n'] == 3 and len(d['windows']['ms_per_step']) == 3
    assert sorted(d['windows']['ms_per_step'])[1] == pytest.approx(d['ms_per_step'], abs=2e-3)
    assert d['value'] == pytest.approx(n * K / (d['ms_per_step'] * 1e-3) / 1e6, rel=2e-3)
    r = d['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'algorithmic_bytes', 'avg_launch_us', 'launches_timed'):
        assert key in r, key
    assert r['bound'] == 'hbm' and r['peak'] == 8000.0 and r['unit'] == 'GB/s' and r['launches_timed'] > 0
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], abs=1e-3)
    assert r['achieved'] == pytest.approx(r['algorithmic_bytes'] / (r['avg_launch_us'] * 1e-6) / 1e9, rel=1e-3)
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == 1 and c['unit'] == 'Mcell-updates/s' and c['value'] > 0 and c['sample']
    assert d['solver']['max_rel_residual'] <= 1e-12


def test_bench_starts_its_own_ranks_for_gpus_2(gpu_lib):
    """`python bench.py --gpus 2` from a plain command line (no launcher, WORLD_SIZE unset): the parent starts the ranks, which
    meet over gloo and exchange halos through the stand-in (both on this one GPU: CWR_BENCH_DEVICE)."""
    from test_gpu_multirank import build_mock
    env = dict(os.environ, CWR_RCCL_LIB=build_mock(), CWR_BENCH_DEVICE='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    d = _line(['--gpus', '2', '--nx', '300', '--ny', '300', '--steps', '3', '--warmup', '2', '--windows', '2'], env=env)
    assert d['n_gpus'] == 2 and d['cpu_baseline'] is None and d['value'] > 0
    assert 'x2' in d['config']['partition']
    it = d['solver']['iterations_per_step']
    assert all(i['exchanges'] > 0 and i['checks'] >= 1 for i in it)
    # the per-rank report that lets a scaling run explain itself
    assert len(d['ranks']) == 2 and [r['rank'] for r in d['ranks']] == [0, 1]
    for r in d['ranks']:
        for key in ('owned_rows', 'computed_rows', 'halo_rows', 'peers', 'passes_per_step', 'mean_pass_us', 'exchanges_alone', 'exchanges_alone_us',
                    'exchanges_beside_compute', 'exchanges_beside_compute_us', 'allreduces', 'allreduces_us', 'checks', 'check_host_wait_us',
                    'standalone_ms_per_step', 'timed_region_ms_per_step'):
            assert key in r, key
        assert r['computed_rows'] >= r['owned_rows'] > 0 and r['peers'] == 1
        assert r['exchanges_alone'] + r['exchanges_beside_compute'] > 0 and r['allreduces'] >= r['checks'] >= 3
        assert r['exchanges_alone_us'] + r['exchanges_beside_compute_us'] > 0 and r['allreduces_us'] > 0 and r['check_host_wait_us'] > 0
        assert r['standalone_ms_per_step'] and r['standalone_ms_per_step'] > 0, r
    assert sum(r['owned_rows'] for r in d['ranks']) == d['config']['cells']
    assert d['compute_side_ceiling']['slowest_rank_standalone_ms_per_step'] == max(r['standalone_ms_per_step'] for r in d['ranks'])


def test_bench_constituent_groups_times_cell_ranges(gpu_lib):
    """`--gpus 4 --k-groups 2`: two cell ranges x two groups of constituents (distributed.GroupedTransport) -- a group is a partitioned run
    of its constituents with a communicator of its own, groups never exchange.  Through the stand-in on the one GPU: same contract line, the
    per-rank report says who carried what."""
    from test_gpu_multirank import build_mock
    env = dict(os.environ, CWR_RCCL_LIB=build_mock(), CWR_BENCH_DEVICE='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    d = _line(['--gpus', '4', '--k-groups', '2', '--constituents', '8', '--nx', '300', '--ny', '300', '--steps', '3', '--warmup', '2', '--windows', '2',
               '--no-rank-ceiling'], env=env)
    assert d['n_gpus'] == 4 and d['value'] > 0 and d['config']['constituents'] == 8
    assert 'x2 x 2 groups of 4 constituents' in d['config']['partition']
    assert [(r['constituent_group'], r['cell_range'], r['constituents']) for r in d['ranks']] == [(0, 0, 4), (0, 1, 4), (1, 0, 4), (1, 1, 4)]
    assert sum(r['owned_rows'] for r in d['ranks']) == 2 * d['config']['cells'] and all(r['peers'] == 1 for r in d['ranks'])
    assert d['solver']['max_rel_residual'] <= 1e-12
