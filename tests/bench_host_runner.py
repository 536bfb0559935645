"""Runs bench.py's main() with the CPU checker engine injected (tests/host_engine.py), so that the
launcher, the rank plumbing and the guarded multi-rank phases of the benchmark script can be
exercised where no GPU exists.  TEST INFRASTRUCTURE: bench.py itself never selects this engine."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from mpi4py_fft_amd import _lib
    from tests.host_engine import HostEngine
    _lib.set_engine(HostEngine())
    import bench
    bench.main()


if __name__ == '__main__':
    main()
