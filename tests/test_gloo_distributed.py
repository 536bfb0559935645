"""world_size > 1 over torch.distributed/gloo on CPU: the N>1 host path (communicator layer,
exchange plans, uneven all-to-all) under the product classes.  See tests/gloo_worker.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('nranks,port', [(2, 29541), (4, 29542), (8, 29543)])
def test_pfft_over_gloo(nranks, port):
    env = dict(os.environ, OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nranks),
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'tests', 'gloo_worker.py')]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert res.returncode == 0 and 'GLOO_WORKER_OK ranks=%d' % nranks in res.stdout, res.stdout[-3000:]
