"""The N>1 path on CPU: two ranks (gloo) each own half of the chains, exchange
the new Z rows after every refresh, and must reproduce the single-rank run bit
for bit (the draws are keyed by global chain id)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import mc3_amd
from mc3_amd import utils as mu
from oracle_engine import OracleChains
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the first `import torch` in a fresh container can take minutes in every worker
pytestmark = pytest.mark.timeout(1200)


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('sampler,thinning', [('demc', 1), ('snooker', 2)])
def test_two_ranks_reproduce_single_rank(tmp_path, sampler, thinning):
    out_path = str(tmp_path / 'shard')
    env = dict(os.environ, OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()),
           os.path.join(ROOT, 'tests', 'dist_worker.py'), out_path, sampler,
           str(thinning)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1000)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]

    x, data, uncert = H.tutorial_data(100)
    np.random.seed(100)
    single = mc3_amd.sample(
        data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
        indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
        pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
        sampler=sampler, nsamples=8*120*thinning, burnin=10, nchains=8,
        thinning=thinning, grtest=False, log=mu.Log(verb=0),
        engine_factory=OracleChains, seed=99)
    for rank in (0, 1):
        got = np.load(f'{out_path}.rank{rank}.npz')
        for key in ('posterior', 'zchain', 'log_post', 'chisq', 'bestp',
                    'acceptance_rate', 'best_log_post', 'medianp', 'stdp'):
            np.testing.assert_array_equal(got[key], single[key], err_msg=key)


def test_communicator_mode_selection_single_process():
    """Mode selection logic without a process group: a single rank never sets up
    an exchange; the partition is the whole ensemble."""
    from mc3_amd import parallel
    comm = parallel.SingleRank()
    assert comm.partition(12) == (0, 12)
    assert comm.mode == 'single' and not comm.wants_p2p and not comm.native
    assert comm.broadcast({'a': 1}) == {'a': 1}
    state = {'numaccept': np.arange(3)}
    assert comm.reduce_state(state) is state


def test_peer_to_peer_falls_back_together():
    """If any rank cannot export its replica, every rank drops the peer-to-peer
    mode (no rank may end up alone in it)."""
    env = dict(os.environ, OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()),
           os.path.join(ROOT, 'tests', 'p2p_fallback_worker.py')]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1000)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count('fallback ok') == 2


def test_peer_to_peer_recovery_logic_on_cpu(tmp_path):
    """One rank reports an aborted peer-to-peer launch: the ranks vote at the next report
    point, BOTH roll back to their checkpoints and redo the lost refreshes through the
    host-staged exchange; the run ends identical to the undisturbed single-rank run
    (TorchDistComm.quiesce / recover; the GPU form is tests/test_gpu_p2p.py)."""
    out_path = str(tmp_path / 'recover')
    env = dict(os.environ, OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()),
           os.path.join(ROOT, 'tests', 'p2p_recover_worker.py'), out_path]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1000)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count('recovered') == 2
    x, data, uncert = H.tutorial_data(100)
    single = mc3_amd.sample(
        data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
        indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
        pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
        sampler='demc', nsamples=8*120, burnin=10, nchains=8, grtest=True,
        log=mu.Log(verb=0), engine_factory=OracleChains, seed=99)
    for rank in (0, 1):
        got = np.load(f'{out_path}.rank{rank}.npz')
        for key in ('posterior', 'zchain', 'log_post', 'chisq', 'bestp',
                    'acceptance_rate', 'best_log_post', 'medianp', 'stdp'):
            np.testing.assert_array_equal(got[key], single[key], err_msg=key)
