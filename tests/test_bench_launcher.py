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
    """ONE attempt (round 2 allowed three: the launcher used to pick a port, release it and let
    torchrun re-bind it, and ranks left through the deadline each on their own; now the rendezvous
    is torchrun's stand-alone one and every rank leaves through bench.Guard, rank 0 last --
    profiles/r03_launcher_loop.txt has the 30-in-a-row log)."""
    return subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)


def _check_line(out, nranks, size):
    assert out['n_gpus'] == nranks and out['steps'] == 2 and out['warmup'] == 1
    assert out['metric'] == 'pfft_3d_c2c_%dcubed_fp64_gflops' % size and out['unit'] == 'GFLOP/s'
    assert out['scaling'] == 'strong' and out['higher_is_better'] is True and out['dtype'] == 'f64'
    assert out['value'] > 0 and out['ms_per_step'] > 0
    assert out['config']['round_trip_rel_err'] <= 1e-10
    # the admission gates of the plan that was timed (tests/test_pencil.py:26-56 of the reference)
    assert out['config']['exchange_check'] == 'bit-exact' and out['config']['forward_rel_err'] <= 2e-10
    assert out['config']['gates']['exchange_hops'] > 0
    assert 'extras_error' not in out, out['extras_error']


def test_bench_self_launch_two_ranks_gloo():
    out = _run(os.path.join(ROOT, 'tests', 'bench_host_runner.py'), 2, 32)
    _check_line(out, 2, 32)
    assert out['config']['grid'] == [2, 1, 1]
    assert [e['ranks'] for e in out['config']['exchange']] == [2]
    assert out['config']['exchange'][0]['route'] == 'direct'
    labels = [l for l, _ in out['stages_ms']['forward']]
    # (slab grid: the two local stages run as one launch, then the exchange, then the far stage)
    assert sum('exchange' in l for l in labels) == 1 and sum(l.startswith('fft') for l in labels) == 2
    assert 'one launch' in labels[0] and 'of 8 TB/s' in labels[0]
    # the self-explaining keys of a multi-GPU run: per-exchange fabric fraction, compute / wire split, overlap
    assert all('xgmi_frac' in l for l in labels if 'exchange' in l)
    ov = out['overlap']
    assert ov['compute_ms'] > 0 and ov['wire_ms'] > 0 and ov['step_ms'] > 0 and ov['overlap_eff'] is not None
    assert out['config']['gates']['forward_rel_err_rounding_ok'] is True
    assert all('overlap' in a for a in out.get('alternatives', []) if 'ms_per_step' in a)


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


def test_bench_drops_a_route_that_misplaces_a_relay_piece():
    """The relayed route with one piece per rank delivered to a neighbouring piece's offset and fetched back from
    there (bench.misroute): forward -> backward still returns the input, so the round trip of rounds 1-4 would have
    admitted the route to timing.  The positional check rejects it, names rank and block, and it is never timed."""
    out = _run(os.path.join(ROOT, 'tests', 'bench_host_runner.py'), 4, 32,
               extra_env={'GFFT_RELAY': 'relay', 'GFFT_RELAY_MIN_BYTES': '0', 'GFFT_BENCH_TEST_MISROUTE': 'measured routes'},
               extra_args=('--no-slab',))
    _check_line(out, 4, 32)
    rm = [a for a in out['alternatives'] if a['plan'] == 'measured routes'][0]
    assert rm['test_misroute_applied'] is True and all(e['route'] == 'relay' for e in rm['exchange'])
    assert rm['exchange_check'] == 'FAILED' and 'misplaced' in rm['rejected'] and 'rank' in rm['exchange_failures'][0]
    assert 'ms_per_step' not in rm and 'value' not in rm
    assert out['config'].get('plan') != 'measured routes' and out['config']['exchange_check'] == 'bit-exact'
    # the same route, intact, is admitted and timed
    ok = _run(os.path.join(ROOT, 'tests', 'bench_host_runner.py'), 4, 32,
              extra_env={'GFFT_RELAY': 'relay', 'GFFT_RELAY_MIN_BYTES': '0'}, extra_args=('--no-slab',))
    rm = [a for a in ok['alternatives'] if a['plan'] == 'measured routes'][0]
    assert rm['exchange_check'] == 'bit-exact' and rm['forward_rel_err'] <= 2e-10 and rm['forward_bit_identical_to_headline']
    assert rm['ms_per_step'] > 0


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


def test_bench_prints_a_diagnostic_line_when_the_headline_fails_on_one_rank():
    """Rank 1 raises before the headline exists while rank 0 is inside the first collective: rank 0
    still prints ONE line -- no value, the error, the wire's diagnostics -- and the job exits 0."""
    env = dict(os.environ, GFFT_DIST_BACKEND='gloo', OMP_NUM_THREADS='1', GFFT_BENCH_TEST_FAIL='headline:1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'tests', 'bench_host_runner.py'), '--gpus', '2', '--size', '16',
           '--no-cpu', '--steps', '1', '--warmup', '0']
    res = _launch(cmd, env, 300)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['value'] is None and out['n_gpus'] == 2 and out['metric'] == 'pfft_3d_c2c_16cubed_fp64_gflops'
    assert 'rank 1' in out['error'] and 'injected failure' in out['error']
    d = out['diagnostics']
    assert 'env' in d and d['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0' and 'torch' in d


def test_bench_headline_deadline_prints_a_line():
    """Initialisation / headline that never finishes (a hung RCCL bootstrap): the line still comes."""
    env = dict(os.environ, GFFT_DIST_BACKEND='gloo', OMP_NUM_THREADS='1', GFFT_BENCH_TEST_HANG='headline')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'tests', 'bench_host_runner.py'), '--gpus', '2', '--size', '16',
           '--no-cpu', '--steps', '1', '--warmup', '0', '--headline-deadline', '15']
    res = _launch(cmd, env, 300)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['value'] is None and 'deadline of 15 s' in out['error'] and "'headline'" in out['error']


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
