"""Physics known-answer test reachable through the path (examples/spectral_dns_solver.py:129 of the
reference): Taylor-Green vortex, 64^3, RK4 x 10 steps -> kinetic energy 0.124953117517 to 7
decimals.  Runs the device port of that example on 1 rank and on 2 / 4 thread-ranks."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))


@pytest.mark.parametrize('P', [1, 2, 4])
def test_taylor_green_energy(P):
    from dns_taylor_green import solve
    from tests import cases
    energies = cases.run_ranks(P, lambda comm: solve(comm))
    for e in energies:
        assert round(e - 0.124953117517, 7) == 0, e
