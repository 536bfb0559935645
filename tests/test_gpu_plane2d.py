"""Small planes on chip (csrc/fft_plane2d.hip): rows and columns of 32 x 32 / 64 x 64 planes in one launch, the plane held in LDS --
BASELINE config C1 (PFFT 64^3 complex128, the size /root/reference/tests/test_speed.py:15-20 times) and fftn(axes=(1, 2)) over
small images.  Parity: numpy on the same seeded input at rounding level, both directions, in place, against the two-pass form."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import cases


def _plans(shape, axes, dt):
    from mpi4py_fft_amd import fftw, zeros
    a = zeros(shape, dt)
    f = fftw.fftn(a, axes=axes)
    b = fftw.ifftn(f.output_array, axes=axes, output_array=zeros(shape, dt))
    return a, f, b


@pytest.mark.parametrize('dt', ['D', 'F'])
@pytest.mark.parametrize('shape,axes', [((64, 64, 64), (0, 1, 2)), ((32, 32, 32), (0, 1, 2)), ((5, 64, 64), (1, 2)), ((1, 32, 32), (1, 2)),
                                        ((3000, 32, 32), (1, 2)), ((1100, 64, 64), (1, 2)), ((48, 64, 64), (0, 1, 2)), ((16, 32, 32), (0, 1, 2)),
                                        ((64, 64), (0, 1)), ((32, 32), (0, 1))])
def test_planes_on_chip_against_numpy(shape, axes, dt):
    from mpi4py_fft_amd import _lib
    rng = np.random.default_rng(17)
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dt)
    npts = int(np.prod([shape[a] for a in axes]))
    ref = np.fft.fftn(x.astype('D'), axes=axes)
    a, f, b = _plans(shape, axes, dt)
    desc = _lib.engine().plan_describe(f._plan)
    assert 'planes on chip' in desc and desc.count('\n') == len(axes), desc          # one launch for the two in-plane axes
    a[...] = x
    got = np.asarray(f.execute_scaled(a, f.output_array, 1.0)).copy()
    cases.assert_close(got, ref, dt, npts, (shape, axes, dt))
    assert np.array_equal(np.asarray(a), x)                                           # out of place: input preserved
    back = np.asarray(b.execute_scaled(f.output_array, b.output_array, 1.0 / npts)).copy()
    cases.assert_roundtrip(back, x, dt, npts, (shape, axes, dt))
    # in place (allowed for complex plans, include/gfft.h)
    f.execute_scaled(a, a, 1.0)
    assert np.array_equal(np.asarray(a), got)
    # ... and the two-pass form of the same plan: same transform, rounding apart
    _lib.set_option('plane2d', 0)
    try:
        a2, f2, b2 = _plans(shape, axes, dt)
        assert 'planes on chip' not in _lib.engine().plan_describe(f2._plan)
        a2[...] = x
        two = np.asarray(f2.execute_scaled(a2, f2.output_array, 1.0))
        assert np.abs(two - got).max() <= cases.rounding_tol(dt, npts) * np.abs(ref).max()
    finally:
        _lib.set_option('plane2d', 1)
    for p in (f, b, f2, b2):
        p.destroy()


def test_c1_runs_two_launches_and_matches_the_oracle():
    """BASELINE config C1 through the public API: PFFT 64^3 complex128 on one rank."""
    from mpi4py_fft_amd import PFFT, comm
    fft = PFFT(comm.COMM_SELF, (64, 64, 64), dtype='D')
    desc = fft._fused_plans[0]._eng.plan_describe(fft._fused_plans[0]._plan)
    assert '2 passes' in desc and 'planes on chip' in desc, desc
    fft.destroy()
    cases.check_pfft_vs_oracle(1, (64, 64, 64), 'D')
    cases.check_pfft_vs_oracle(1, (64, 64, 64), 'F')
    cases.check_pfft_vs_oracle(1, (32, 32, 32), 'D')
