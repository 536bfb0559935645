"""The C ABI used from plain C (examples/capi_demo.c): compiled with gcc against include/gfft.h,
linked to libgfft.so, run on the GPU; its own parity checks decide the exit code."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_demo_from_c(tmp_path):
    exe = str(tmp_path / 'capi_demo')
    lib = os.path.join(ROOT, 'mpi4py-fft_amd')
    subprocess.check_call(['gcc', '-O2', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'examples', 'capi_demo.c'), '-o', exe,
                           '-L', lib, '-lgfft', '-lm', '-Wl,-rpath,' + lib])
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0 and 'capi_demo OK' in res.stdout, res.stdout
    # the GPU box HAS an RCCL: the exchange section must have run on it, not printed its 'skipped' line
    assert 'exchange: RCCL bound from' in res.stdout and 'skipped' not in res.stdout, res.stdout
