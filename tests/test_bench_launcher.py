"""`python bench.py --gpus N` from a plain shell: self-launch under torch.distributed.run, one JSON
line from rank 0, guarded multi-rank phases.  CPU: gloo + the checker engine through
tests/bench_host_runner.py; GPU: the real script, two processes sharing the one GPU over gloo
(RCCL refuses two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, nranks, size, extra_env=None, extra_args=()):
    env = dict(os.environ, GFFT_DIST_BACKEND='gloo', OMP_NUM_THREADS='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(extra_env or {})
    cmd = [sys.executable, script, '--gpus', str(nranks), '--size', str(size), '--no-cpu', '--steps', '2',
           '--warmup', '1'] + list(extra_args)
    res = _launch(cmd, env, 600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def _launch(cmd, env, timeout):
    """Up to three attempts: the launcher picks a free rendezvous port and releases it before torchrun
    binds it, which another process on a busy test box can win; and a rank that leaves through the
    deadline (os._exit) while its peer is still inside a gloo call has been seen to take the peer
    down with a connection error about once in thirty runs.  A run counts when it exits 0 AND rank 0
    printed its one line."""
    for attempt in range(3):
        res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
        if res.returncode == 0 and sum(l.startswith('{') for l in res.stdout.splitlines()) == 1:
            break
    return res


def _check_line(out, nranks, size):
    assert out['n_gpus'] == nranks and out['steps'] == 2 and out['warmup'] == 1
    assert out['metric'] == 'pfft_3d_c2c_%dcubed_fp64_gflops' % size and out['unit'] == 'GFLOP/s'
    assert out['scaling'] == 'strong' and out['higher_is_better'] is True and out['dtype'] == 'f64'
    assert out['value'] > 0 and out['ms_per_step'] > 0
    assert out['config']['round_trip_rel_err'] <= 1e-10
    assert 'extras_error' not in out, out['extras_error']


def test_bench_self_launch_two_ranks_gloo():
    out = _run(os.path.join(ROOT, 'tests', 'bench_host_runner.py'), 2, 32)
    _check_line(out, 2, 32)
    assert out['config']['grid'] == [2, 1, 1]
    assert [e['ranks'] for e in out['config']['exchange']] == [2]
    assert out['config']['exchange'][0]['route'] == 'direct'
    labels = [l for l, _ in out['stages_ms']['forward']]
    assert sum('exchange' in l for l in labels) == 1 and sum(l.startswith('fft') for l in labels) == 3


def test_bench_self_launch_four_ranks_measures_routes_and_slab():
    out = _run(os.path.join(ROOT, 'tests', 'bench_host_runner.py'), 4, 32,
               extra_env={'GFFT_RELAY': 'measure', 'GFFT_RELAY_MIN_BYTES': '0'})
    _check_line(out, 4, 32)
    assert out['config']['grid'] == [2, 2, 1]
    rm = [a for a in out['alternatives'] if a['plan'] == 'measured routes'][0]
    assert any(a['plan'] == 'pipelined' for a in out['alternatives'])
    assert rm["round_trip_rel_err"] <= 1e-10
    assert all(len(e['measured_s']) == 2 and e['route'] in ('direct', 'relay') for e in rm['exchange'])
    assert out['slab_grid']['grid'] == [4, 1, 1] and out['slab_grid']['gflops'] > 0


def test_bench_keeps_the_headline_when_an_extra_hangs():
    """A phase after the headline that never returns: the deadline prints the line anyway."""
    out_lines = None
    env = dict(os.environ, GFFT_DIST_BACKEND='gloo', OMP_NUM_THREADS='1', GFFT_BENCH_TEST_HANG='stage breakdown')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'tests', 'bench_host_runner.py'), '--gpus', '2', '--size', '16',
           '--no-cpu', '--steps', '1', '--warmup', '0', '--extras-deadline', '5']
    res = _launch(cmd, env, 300)
    assert res.returncode == 0, res.stderr[-3000:]
    out_lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(out_lines) == 1
    out = json.loads(out_lines[0])
    assert out['value'] > 0 and 'deadline' in out['extras_error']


@pytest.mark.gpu
def test_bench_self_launch_two_processes_share_the_gpu():
    out = _run(os.path.join(ROOT, 'bench.py'), 2, 128)
    _check_line(out, 2, 128)
    assert out['config']['grid'] == [2, 1, 1]
    assert out['roofline']['bound'] == 'hbm' and out['roofline']['achieved'] > 0
    assert any('exchange' in l for l, _ in out['stages_ms']['forward'])


@pytest.mark.gpu
def test_bench_single_gpu_line_small():
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--size', '128', '--steps', '3', '--warmup', '1',
                          '--no-cpu'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][0])
    assert out['n_gpus'] == 1 and out['roofline']['frac'] > 0 and out['config']['grid'] == [1, 1, 1]
