"""Rounding-level parity guards (round 6).

The contract tolerances of the north star (fp64 2e-10 / 1e-10, fp32 2e-4 / 1e-4) are 10^3 ... 10^5 times looser than the
accuracy the kernels have: a kernel that lost three digits in fp32, or five in a distributed fp64 plan, would pass every
comparison written against them.  Two guards sit beside them since round 6:

  * `tests/cases.py rounding_tol` -- 2 eps log2(points) -- asserted wherever a forward or a round trip is compared (the
    shared checkers `check_pfft_golden`, `check_pfft_vs_oracle`, `assert_close`, `assert_roundtrip`, the serial `_check`
    of test_gpu_serial.py, the full-size DFT lines of test_gpu_large.py / test_gpu_c5.py, test_gpu_fused2.py);
  * impulse responses (`tests/cases.py impulse_errors`): the transform of a unit impulse is a bare product of the plan's
    twiddle factors, of modulus 1, so a wrong table entry shows at full size (random data dilutes it by ~1 / (3.5 sqrt(radix))).

This file holds the impulse family over every kernel family and both precisions, and the proof that the guards bite: one
twiddle entry off by 1e-9 (fp64) / 1e-5 (fp32), injected through the library's test hook (`gfft_set_option
"debug_tw_exp"`, plan.cpp get_twiddles), fails them while the contract tolerances alone let it through.
Reference tolerance for the same kind of check: /root/reference/tests/test_libfft.py:17 (abstol f = 5e-5, d = 1e-14).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import cases

# power-of-two register kernels; 3^b 2^k; 5^c 2^k; unequal-width stages (3 x 5 x 2^k, 7 x 2^k); generic LDS kernel; four-step; Bluestein
LENGTHS = [16, 64, 256, 512, 1024, 2048, 4096, 48, 384, 768, 3072, 20, 80, 640, 1000, 4000, 240, 960, 112, 896, 1792,
           22, 34, 121, 343, 1331, 8192, 16384, 9216, 521, 1009]
IMPULSE_FACTOR = 2.0          # measured (tools/impulse_probe.py, profiles/r06_impulse_probe.txt): <= 1.2 (fp64), <= 0.51 (fp32) eps log2 n


@pytest.mark.parametrize('strided', [False, True], ids=['rows', 'strided'])
@pytest.mark.parametrize('dt', ['D', 'F'])
def test_impulse_responses_at_rounding_level(dt, strided):
    worst = 0.0
    for n in LENGTHS:
        ef, eb = cases.impulse_errors(n, dt, strided)
        unit = cases.EPS[dt] * np.log2(n)
        worst = max(worst, ef / unit, eb / unit)
        assert ef <= IMPULSE_FACTOR * unit, (n, dt, strided, 'forward', ef, ef / unit)
        assert eb <= IMPULSE_FACTOR * unit, (n, dt, strided, 'backward', eb, eb / unit)
    assert worst > 0          # (the transforms did run)


@pytest.fixture
def wrong_twiddle():
    from mpi4py_fft_amd import _lib

    def arm(index, exponent):
        _lib.set_option('debug_tw_index', index)
        _lib.set_option('debug_tw_exp', exponent)
    yield arm
    _lib.set_option('debug_tw_exp', 0)
    _lib.set_option('debug_tw_index', 1)


@pytest.mark.parametrize('dt,exponent', [('D', 9), ('D', 11), ('F', 5)])
def test_one_wrong_twiddle_entry_fails_the_guards(dt, exponent, wrong_twiddle):
    """Plans made while the hook is armed read tables whose entry 1 (w^1: every plan's last stage reads it) is off by
    10^-exponent in its real part.  The impulse family must name it in both precisions, in both mappings, at every
    length; the random-data guards must name it in fp64 (1e-9 and 1e-11 against ~1e-14).  The CONTRACT tolerances let the
    fp64 entry off by 1e-11 and the fp32 one off by 1e-5 through (on 64^3 the 1e-9 one comes out at 1.3e-9, which they
    catch) -- which is why they cannot be the only check."""
    delta = 10.0 ** -exponent
    shape = (64, 64, 64)
    cases.check_pfft_vs_oracle(1, shape, dt)                      # sane before ...
    wrong_twiddle(1, exponent)
    for n in (16, 64, 512, 1024, 2048, 768, 640, 960):
        for strided in (False, True):
            ef, eb = cases.impulse_errors(n, dt, strided)
            unit = cases.EPS[dt] * np.log2(n)
            assert ef > IMPULSE_FACTOR * unit and ef >= 0.9 * delta, (n, dt, strided, ef)          # caught, at full size
    # a whole transform on random data: what the shared checker sees
    from mpi4py_fft_amd import PFFT, newDistArray, comm
    from oracle import pfft_oracle as O
    fft = PFFT(comm.COMM_SELF, shape, dtype=dt)
    G = O.rng_array(shape, dt, 7)
    u = newDistArray(fft, False)
    u[...] = G
    uh = np.asarray(fft.forward(u)).copy()
    ref = O.OPFFT(1, shape, dtype=dt).forward([G])[0]
    err = float(np.abs(uh - ref).max() / np.abs(ref).max())
    fft.destroy()
    if exponent != 9:
        assert err <= cases.CONTRACT['fwd'][dt], err              # the contract tolerance passes the broken table
    if dt == 'D':
        assert err > cases.rounding_tol(dt, G.size), (err, cases.rounding_tol(dt, G.size))
        with pytest.raises(AssertionError, match='ROUNDING-LEVEL' if exponent != 9 else None):
            cases.check_pfft_vs_oracle(1, shape, dt)
    else:
        # fp32: one entry off by 1e-5 is diluted to ~1 / (3.5 sqrt(radix)) of itself on random data -- level with fp32
        # rounding; the impulse family above is what names it
        assert err <= 10 * cases.rounding_tol(dt, G.size)
    wrong_twiddle(1, 0)
    cases.check_pfft_vs_oracle(1, shape, dt)                      # ... and sane after: plans made now read the exact tables


def test_lost_digits_in_a_distributed_plan_fail_the_guards(wrong_twiddle):
    """The same on 4 thread-ranks (pencil grid, packed exchange buffers): a distributed fp64 plan that lost three digits
    passes the 2e-10 contract and fails the shared checker."""
    shape = (64, 32, 64)
    cases.check_pfft_vs_oracle(4, shape, 'D')
    wrong_twiddle(1, 11)
    with pytest.raises(AssertionError, match='ROUNDING-LEVEL'):
        cases.check_pfft_vs_oracle(4, shape, 'D')
    wrong_twiddle(1, 0)
    cases.check_pfft_vs_oracle(4, shape, 'D')
