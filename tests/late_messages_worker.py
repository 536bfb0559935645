"""Run by tests/test_gpu_pipeline.py in a process of its own with FAKE_RCCL_DELAY_US set: every
message of the stand-in wire lands late, so a transform stage that does not wait for the arrival
event of its chunk reads stale data.  `broken` removes those waits on purpose (negative control: the
harness must notice)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpi4py_fft_amd import _lib, pipeline          # noqa: E402
from tests import cases                            # noqa: E402

so = os.path.join(ROOT, 'tests', 'fake_rccl', 'libfake_rccl.so')
_lib.check_wire(_lib.lib().gfft_rccl_load(so.encode()))
pipeline.Pipeline.MIN_CHUNK_BYTES = 0
pipeline.Pipeline.MIN_WIDTH = 4
os.environ['GFFT_WIRE'] = 'native'
broken = len(sys.argv) > 1 and sys.argv[1] == 'broken'
if broken:
    pipeline.Pipeline._arrived = lambda self, *a: None
try:
    for relay in ('0', '1'):
        os.environ['GFFT_RELAY'] = relay
        for chunks in (2, 4):
            pipeline.Pipeline.CHUNKS = chunks
            cases.check_pfft_vs_oracle(8, (64, 64, 128), 'D')
            cases.check_pfft_vs_oracle(8, (64, 64, 128), 'f')
            cases.check_pfft_vs_oracle(4, (64, 128, 64), 'F', grid=(-1,))
            cases.check_pfft_vs_oracle(4, (32, 64, 64), 'd')
    # two transforms back to back on different inputs: the second one's first stage must not
    # overwrite exchange buffers the first one's messages are still being copied out of
    from mpi4py_fft_amd import PFFT, newDistArray
    from oracle import pfft_oracle as O
    shape = (64, 64, 128)
    G1, G2 = O.rng_array(shape, 'D', 1), O.rng_array(shape, 'D', 2)

    def body(comm):
        f = PFFT(comm, shape, dtype='D', wire='native', exchange='direct')
        assert f.pipeline is not None
        u1, u2 = newDistArray(f, False), newDistArray(f, False)
        o1, o2 = newDistArray(f, True), newDistArray(f, True)
        u1[...] = G1[f.local_slice(False)]
        u2[...] = G2[f.local_slice(False)]
        f.forward(u1, o1)
        f.forward(u2, o2)
        b1, b2 = newDistArray(f, False), newDistArray(f, False)
        f.backward(o1, b1)
        f.backward(o2, b2)
        res = [np.asarray(x).copy() for x in (o1, o2, b1, b2)]
        sl = f.local_slice(False)
        f.destroy()
        return res, sl
    ref = O.OPFFT(8, shape, dtype='D')
    w1, w2 = ref.forward(ref.scatter(G1)), ref.forward(ref.scatter(G2))
    for r, ((o1, o2, b1, b2), sl) in enumerate(cases.run_ranks(8, body)):
        assert np.abs(o1 - w1[r]).max() <= 1e-12 * np.abs(w1[r]).max()
        assert np.abs(o2 - w2[r]).max() <= 1e-12 * np.abs(w2[r]).max()
        assert np.allclose(b1, G1[sl], rtol=0, atol=1e-12) and np.allclose(b2, G2[sl], rtol=0, atol=1e-12)
    print('late-messages: results correct')
except AssertionError:
    print('late-messages: results WRONG')
    sys.exit(0 if broken else 1)
sys.exit(1 if broken else 0)
