"""bench.py's cpu_baseline leg: the sample ladder must reach 512^3 whatever the planner costs
(round 2's record stopped at a 128^3 in-cache sample because 5 s of planning ate the budget)."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class _Clock:
    def __init__(self):
        self.t = 0.0

    def __call__(self):
        return self.t


class _FakeFftw:
    """Stands in for bench._Fftw: plans cost `plan_s` each on the injected clock, an execute costs
    `exec_s_256` scaled with the cube's size, and does nothing (fwd then bwd of nothing is the
    identity, so the round-trip check of the ladder holds)."""
    MEASURE, ESTIMATE = 0, 64
    kind, path, version, threads = 'fake', '/fake/libfftw3.so', 'fake FFTW', 3

    def __init__(self, clock, plan_s, exec_s_256):
        self.clock, self.plan_s, self.exec_s_256 = clock, plan_s, exec_s_256
        self.n = None
        self.planned, self.flags = [], set()
        self.lib = types.SimpleNamespace(fftw_execute_dft=self._execute, fftw_destroy_plan=lambda p: None)

    def plan(self, arr_in, arr_out, axes, sign, flags):
        self.clock.t += self.plan_s
        self.n = arr_in.shape[0]
        self.planned.append((arr_in.shape, tuple(axes), sign))
        self.flags.add(flags)
        return len(self.planned)

    def _execute(self, plan, a, b):
        self.clock.t += self.exec_s_256 * (self.n / 256.0) ** 3 / 3.0


def test_a_slow_planner_cannot_stop_the_ladder_below_512():
    clock = _Clock()
    F = _FakeFftw(clock, plan_s=6.0, exec_s_256=0.05)
    out = bench._cpu_fftw(8, 25.0, F=F, clock=clock, max_n=512)
    assert out['sample'].startswith('512^3 complex128 fwd+bwd, best of '), out
    assert 'after one warm-up' in out['sample'] and 'planning 36.0 s' in out['sample'], out
    n_runs = int(out['sample'].split('best of ')[1].split()[0])
    assert n_runs >= 3
    assert out['cores'] == 3 and out['kind'] == 'port' and out['unit'] == 'GFLOP/s'
    assert F.flags == {F.ESTIMATE}                      # no bounded MEASURE on this library: ESTIMATE
    # per-axis plans in the reference's stage order (mpifft.py:313-331), both rungs
    assert [p[1] for p in F.planned[:6]] == [(2,), (1,), (0,), (0,), (1,), (2,)]
    assert {p[0] for p in F.planned} == {(256,) * 3, (512,) * 3}


def test_512_is_taken_beyond_the_budget_but_within_the_cap_and_refused_beyond_it():
    clock = _Clock()
    # 256^3 at 2 s per fwd+bwd: 512^3 predicted ~80 s for warm-up + 3 runs: over the 25 s budget,
    # inside the 120 s cap -> taken; the 1024^3 rung is then refused and the line says why
    F = _FakeFftw(clock, plan_s=0.1, exec_s_256=1.0)
    out = bench._cpu_fftw(8, 25.0, F=F, clock=clock, max_n=512)
    assert out['sample'].startswith('512^3'), out
    clock2 = _Clock()
    F2 = _FakeFftw(clock2, plan_s=0.1, exec_s_256=5.0)
    out2 = bench._cpu_fftw(8, 25.0, F=F2, clock=clock2, max_n=512)
    assert out2['sample'].startswith('256^3'), out2
    assert 'ladder stopped: 512^3 predicted at' in out2['sample'] and '120 s cap' in out2['sample'], out2


def test_cpu_baseline_falls_back_to_the_oracle_when_no_fftw_loads(monkeypatch):
    def boom(*a, **k):
        raise OSError('no FFTW3 implementation found')
    monkeypatch.setattr(bench, '_cpu_fftw', boom)
    seen = {}

    def fake_pocket(cores, budget_s):
        seen['args'] = (cores, budget_s)
        return dict(value=1.0, unit='GFLOP/s', cores=cores, kind='port', library='scipy pocketfft', sample='x')
    monkeypatch.setattr(bench, '_cpu_pocketfft', fake_pocket)
    out = bench.cpu_baseline(4, 5.0)
    assert seen['args'] == (4, 5.0) and 'FFTW probe' in out['sample']


def test_per_axis_plans_on_a_one_batch_dim_library_equal_fftn():
    """What the cpu_baseline leg executes when the FFTW3 implementation at hand is MKL's interface (one batch dim per
    guru plan): merged contiguous batch dims for axes 2 and 0, a loop over the slabs of one 2-D plan for axis 1 --
    the reference's default per-axis stage order (mpifft.py:313-331), checked by value, not by a round trip."""
    import numpy as np
    import pytest
    try:
        F = bench._Fftw(2)
    except OSError:
        pytest.skip('no FFTW3 implementation loadable here')
    n = 24
    rng = np.random.default_rng(0)
    u = rng.standard_normal((n, n, n)) + 1j * rng.standard_normal((n, n, n))
    v = np.empty_like(u)
    ex = F.lib.fftw_execute_dft
    p2 = F.plan(u, v, [2], -1, F.ESTIMATE)
    assert p2
    ex(p2, u.ctypes.data, v.ctypes.data)
    p1 = F.plan(v, v, [1], -1, F.ESTIMATE)
    if p1:
        ex(p1, v.ctypes.data, v.ctypes.data)
    else:
        p1 = F.plan(v[0], v[0], [0], -1, F.ESTIMATE)
        assert p1
        for i in range(n):
            ex(p1, v.ctypes.data + i * n * n * 16, v.ctypes.data + i * n * n * 16)
    p0 = F.plan(v, v, [0], -1, F.ESTIMATE)
    assert p0
    ex(p0, v.ctypes.data, v.ctypes.data)
    assert np.abs(v - np.fft.fftn(u)).max() < 1e-11
