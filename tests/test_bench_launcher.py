"""`python bench.py --gpus N` must start its own ranks (CPU test of the launcher; the children are stubs).

The driver's N = 1 command has no launcher in front of it; for N > 1 both shapes have to work: under
torch.distributed.run (WORLD_SIZE set: bench.py is a rank) and plainly (bench.py becomes the parent of N ranks).
The parent never imports torch and never touches the GPU."""
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(tmp_path, child_src, n=3, timeout=60):
    child = tmp_path / 'child.py'
    child.write_text(textwrap.dedent(child_src))
    env = dict(os.environ, CWR_BENCH_CHILD=str(child), CWR_TEST_DIR=str(tmp_path))
    env.pop('WORLD_SIZE', None); env.pop('RANK', None)
    return subprocess.run([sys.executable, BENCH, '--gpus', str(n), '--steps', '4', '--warmup', '2'], env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_launcher_starts_n_ranks_and_relays_rank0_line(tmp_path):
    p = _run(tmp_path, '''
        import json, os, sys
        r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0
        assert int(os.environ['LOCAL_RANK']) == r
        open(os.path.join(os.environ['CWR_TEST_DIR'], f'rank{r}.args'), 'w').write(' '.join(sys.argv[1:]))
        print('chatter that is not the result line')
        if r == 0:
            print(json.dumps({'metric': 'Mcell-updates/s', 'n_gpus': w, 'value': 1.0}))
    ''')
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                          # ONE JSON line on stdout, nothing else
    assert json.loads(lines[0])['n_gpus'] == 3
    for r in range(3):                                        # every rank got the parent's command line
        assert (tmp_path / f'rank{r}.args').read_text() == '--gpus 3 --steps 4 --warmup 2'


def test_a_failing_rank_fails_the_run_and_leaves_no_orphan(tmp_path):
    t0 = time.monotonic()
    p = _run(tmp_path, '''
        import os, sys, time
        r = int(os.environ['RANK'])
        open(os.path.join(os.environ['CWR_TEST_DIR'], f'rank{r}.pid'), 'w').write(str(os.getpid()))
        if r == 1:
            time.sleep(0.5)
            sys.exit(7)
        time.sleep(300)                                       # the others wait "in a collective"
    ''')
    assert p.returncode == 7
    assert p.stdout.strip() == ''                             # no result line from a failed run
    assert time.monotonic() - t0 < 30                         # the waiting ranks were stopped, not waited for
    for r in range(3):
        pid = int((tmp_path / f'rank{r}.pid').read_text())
        assert not os.path.exists(f'/proc/{pid}'), f'rank {r} (pid {pid}) outlived the launcher'


def test_rank0_without_a_result_line_is_a_failure(tmp_path):
    p = _run(tmp_path, 'print("no json here")\n', n=2)
    assert p.returncode != 0
    assert p.stdout.strip() == ''


def test_under_a_launcher_a_wrong_world_size_is_refused():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0')
    p = subprocess.run([sys.executable, BENCH, '--gpus', '2'], env=env, capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and 'WORLD_SIZE' in p.stderr


def test_sigterm_to_the_launcher_stops_the_ranks(tmp_path):
    """the ranks run in sessions of their own; an outer `timeout` (SIGTERM to the launcher) used to skip the reaping
    and leave them waiting in a collective, holding their GPUs."""
    import signal
    child = tmp_path / 'child.py'
    child.write_text(textwrap.dedent('''
        import os, time
        r = int(os.environ['RANK'])
        open(os.path.join(os.environ['CWR_TEST_DIR'], f'rank{r}.pid'), 'w').write(str(os.getpid()))
        time.sleep(300)
    '''))
    env = dict(os.environ, CWR_BENCH_CHILD=str(child), CWR_TEST_DIR=str(tmp_path))
    env.pop('WORLD_SIZE', None); env.pop('RANK', None)
    p = subprocess.Popen([sys.executable, BENCH, '--gpus', '3'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    deadline = time.monotonic() + 30
    while time.monotonic() < deadline and not all((tmp_path / f'rank{r}.pid').exists() for r in range(3)):
        time.sleep(0.1)
    assert all((tmp_path / f'rank{r}.pid').exists() for r in range(3))
    time.sleep(0.2)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=30)
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err)
    assert out.strip() == ''
    for r in range(3):
        pid = int((tmp_path / f'rank{r}.pid').read_text())
        assert not os.path.exists(f'/proc/{pid}'), f'rank {r} (pid {pid}) outlived the launcher'
