"""Randomised cross-check of the planner: random shapes (incl. size-1 and awkward lengths), axis
subsets, kinds and precisions through the C ABI against the oracle.  Seeded, so reproducible."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pfft_oracle as O

POOL = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 16, 17, 20, 24, 25, 27, 30, 32, 36, 45, 48, 49, 50, 60, 64,
        72, 81, 96, 100, 108, 121, 125, 128, 144, 169, 192, 200, 210, 243, 256, 288, 343, 384, 500, 512, 625]


@pytest.mark.parametrize('seed', range(8))
def test_random_serial_plans(seed):
    from mpi4py_fft_amd import FFT, asdevice
    rng = np.random.default_rng(1000 + seed)
    for _ in range(40):
        nd = int(rng.integers(1, 5))
        while True:
            shape = tuple(int(rng.choice(POOL)) for _ in range(nd))
            if np.prod(shape) <= 3_000_000:
                break
        k = int(rng.integers(1, nd + 1))
        axes = tuple(int(a) for a in rng.permutation(nd)[:k])
        dt = str(rng.choice(list('dDfF')))
        if dt in 'df' and shape[axes[-1]] < 2:
            continue
        fft = FFT(shape, axes, dtype=dt)
        ref = O.OFFT(shape, axes, dt)
        A = O.rng_array(shape, dt, int(rng.integers(1 << 30)))
        B = np.asarray(fft.forward(asdevice(A)))
        Bref = ref.forward(A)
        tol = 2e-10 if dt in 'dD' else 3e-4
        assert B.shape == Bref.shape and B.dtype == Bref.dtype, (shape, axes, dt)
        assert np.abs(B - Bref).max() <= tol * max(np.abs(Bref).max(), 1e-30), (shape, axes, dt)
        A2 = np.asarray(fft.backward(asdevice(Bref)))
        assert A2.shape == A.shape
        assert np.linalg.norm(A2 - A) <= (1e-10 if dt in 'dD' else 2e-4) * np.linalg.norm(A), (shape, axes, dt)
        fft.destroy()


@pytest.mark.parametrize('seed', range(4))
def test_random_pfft(seed):
    """Random PFFT configurations on 1-4 thread-ranks vs the oracle."""
    from tests import cases
    rng = np.random.default_rng(2000 + seed)
    pool = [8, 9, 12, 13, 16, 18, 20, 24, 27, 32, 48]
    done = 0
    while done < 10:
        nd = int(rng.integers(2, 5))
        shape = tuple(int(rng.choice(pool)) for _ in range(nd))
        if np.prod(shape) > 400_000:
            continue
        P = int(rng.choice([1, 2, 3, 4]))
        dt = str(rng.choice(list('dDF')))
        kw = {}
        if rng.random() < 0.4:
            kw['collapse'] = True
        if rng.random() < 0.3 and nd >= 3:
            kw['grid'] = (-1,)
        # the reference needs shape[i] >= ranks on distributed axes incl. the halved one
        halved = shape[-1] // 2 + 1 if dt in 'd' else shape[-1]
        if min(shape[:-1] + (halved,)) < P:
            continue
        try:
            cases.check_pfft_vs_oracle(P, shape, dt, seed=int(rng.integers(1 << 30)), **kw)
        except AssertionError as e:
            if 'shape[i] >= size' in str(e):
                continue
            raise
        done += 1


@pytest.mark.parametrize('seed', range(3))
def test_random_redistribute_chains(seed):
    """Random DistArrays (uneven blocks, tensor ranks 0-2, slab ... pencil grids on 2-8 thread ranks)
    walked through random redistributions: every rank holds its local_slice() of the global array
    after each (tests/test_darray.py:57-96 does one fixed walk)."""
    from tests import cases
    rng = np.random.default_rng(3000 + seed)
    done = 0
    while done < 12:
        done += bool(cases.check_redistribute_chain(rng))
    if seed == 0:
        rng = np.random.default_rng(3100)
        done = 0
        while done < 3:
            done += bool(cases.check_redistribute_chain(rng, mid=True))
