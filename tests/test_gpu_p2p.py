"""The cross-GPU peer-to-peer hand-off, exercised with two PROCESSES on one GPU:
each rank owns half of the chains and a full Z replica, opens the other's replica
through HIP IPC, publishes its rows into both and polls its own.  Must reproduce
the single-rank run bit for bit (draws are keyed by global chain id)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import helpers as H

# the first `import torch` on a fresh box can take minutes in every worker process
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1200)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('nproc,sampler,thinning,niter', [
    (2, 'demc', 1, 400), (2, 'snooker', 2, 200), (4, 'demc', 1, 300), (4, 'snooker', 1, 300),
    (8, 'demc', 1, 200)])
def test_processes_peer_to_peer(tmp_path, nproc, sampler, thinning, niter):
    import mc3_amd
    from mc3_amd import utils as mu
    from oracle_engine import OracleChains
    out_path = str(tmp_path / 'p2p')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={nproc}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()),
           os.path.join(ROOT, 'tests', 'p2p_worker.py'), out_path, sampler,
           str(thinning), str(niter)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1000)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-6000:]

    x, data, uncert = H.tutorial_data(100)
    np.random.seed(100)
    single = mc3_amd.sample(
        data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
        indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
        pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
        sampler=sampler, nsamples=16*niter*thinning, burnin=10, nchains=16,
        thinning=thinning, grtest=True, log=mu.Log(verb=0),
        engine_factory=OracleChains, seed=99)
    for rank in range(nproc):
        got = np.load(f'{out_path}.rank{rank}.npz')
        for key in ('posterior', 'zchain', 'log_post', 'chisq', 'bestp',
                    'acceptance_rate', 'best_log_post', 'medianp', 'stdp'):
            np.testing.assert_array_equal(got[key], single[key], err_msg=key)
