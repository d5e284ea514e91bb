"""The node's tests, rehearsed on a box without a GPU.  tests/test_gpu_multidevice.py needs two
GPUs and more, so it has never run anywhere (the development lease has one); the first time it
does must not be lost to a typo.  Here, with a pretended device count of 8:

  * the module is collected (its parametrisation over 2 / 4 / 8 ranks, its imports);
  * the very worker command lines its tests launch -- tests/c3_worker.py with `devices=own`,
    `exchange=rccl` and peer to peer, tests/p2p_worker.py under both launchers -- run as real
    process groups (socket rendezvous, torchrun + gloo) up to their first GPU call: arguments,
    the communicator, engine_options, the partition, the problem, the population file; the
    library then says there is no device, and the ranks leave together with exit code 0.

What the reference does at this point: mc3/mcmc_driver.py:204-221 (the fan-out), :300-307 (the
barrier per iteration)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
import ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(H.gpu_count() > 0 and not os.environ.get('MC3_TEST_FAKE_GPUS'),
                                reason='a rehearsal for boxes without a GPU')


def test_the_nodes_tests_are_collected_with_eight_pretended_gpus():
    env = dict(os.environ, MC3_TEST_FAKE_GPUS='8')
    res = subprocess.run([sys.executable, '-m', 'pytest', '--collect-only', '-q',
                          os.path.join(ROOT, 'tests', 'test_gpu_multidevice.py')],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    ids = [line for line in res.stdout.splitlines() if '::' in line]
    assert any('test_rccl_all_gather_with_a_rank_per_gpu[0-8]' in i for i in ids), ids
    assert any('test_peer_to_peer_hand_off_across_gpus' in i and '-8]' in i for i in ids), ids
    assert any('test_config_c3_sharded_over_the_node' in i for i in ids), ids
    assert len(ids) >= 6 + 15 + 1 + 2, len(ids)


def _population(tmp_path, nchains, nref, **shape):
    prob = H.lean_shape_problem(nchains, nref, **shape)
    orc = H.make_oracle(prob)
    Z0, lp0 = H.initial_population(orc)
    out = str(tmp_path / 'c3')
    np.savez(out + '.pop.npz', Z0=Z0, lp0=lp0)
    return out


@pytest.mark.parametrize('nranks,extra', [
    (8, ['zlag=0', 'per_rank=64', 'devices=own', 'exchange=rccl']),       # the RCCL all-gather
    (2, ['zlag=2', 'per_rank=64', 'devices=own', 'exchange=rccl']),
    (8, ['zlag=0', 'devices=own']),                                       # peer to peer, C3
    (4, ['zlag=4', 'per_rank=96', 'devices=own', 'sampler=snooker']),
    (2, ['zlag=1', 'per_rank=96', 'devices=own', 'sampler=demc', 'thinning=2'])])
def test_engine_level_workers_reach_their_first_gpu_call(tmp_path, nranks, extra):
    kv = dict(a.split('=') for a in extra)
    per_rank = int(kv.get('per_rank', 256))
    out = _population(tmp_path, per_rank*nranks, 12, sampler=kv.get('sampler', 'demc'),
                      thinning=int(kv.get('thinning', 1)))
    text = ranks.launch_ranks(nranks, 'c3_worker.py', out, 12, *extra,
                              env={'MC3_TEST_DRY_RUN': '1'}, timeout=300)
    assert text.count('dry run reached the first GPU call') == nranks, text[-2000:]


@pytest.mark.parametrize('launcher', ['socket', 'torchrun'])
def test_sample_workers_reach_their_first_gpu_call(tmp_path, launcher):
    text = ranks.launch_ranks(2, 'p2p_worker.py', str(tmp_path / 'p2p'), 'demc', 1, 40,
                              'nchains=16', launcher=launcher,
                              env={'MC3_TEST_DRY_RUN': '1', 'P2P_WORKER_DEVICES': 'own'},
                              timeout=300)
    assert text.count('dry run reached the first GPU call') == 2, text[-2000:]
