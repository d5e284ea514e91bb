"""What ONE GPU cannot show (VERDICT r4 row h): the exchanges of a sharded run
with every rank on a GPU OF ITS OWN -- the RCCL all-gather of the north star with more than one
rank, and the peer-to-peer hand-off over real xGMI (HIP IPC attach across devices, remote
stores acknowledged over the link).  Skipped on boxes with one GPU (the development lease);
they light up the moment the suite runs on a node.  Same bar as everywhere: every rank's
replica equals the oracle's run of the whole ensemble, bit for bit.

The workers are tests/c3_worker.py (engine level, `devices=own`, `exchange=rccl`) and
tests/p2p_worker.py (mc3_amd.sample() end to end, `devices=own`)."""
import os

import numpy as np
import pytest

import helpers as H
from test_gpu_p2p import lean_ranks_case, p2p_case

NGPU = H.gpu_count()
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1200),
              pytest.mark.skipif(NGPU < 2, reason=f'needs two GPUs (this box shows {NGPU})')]


def _ranks():
    return [n for n in (2, 4, 8) if n <= NGPU]


@pytest.mark.parametrize('nranks', _ranks() or [2])
@pytest.mark.parametrize('zlag', [0, 2])
def test_rccl_all_gather_with_a_rank_per_gpu(tmp_path, nranks, zlag):
    """Row h: `step_kernel` + ONE in-place ncclAllGather of the refresh's Z rows, replayed from
    a hipGraph, with nranks > 1 -- RCCL reports the rank count, the replicas equal the oracle
    (strict lockstep and a cadence)."""
    lean_ranks_case(tmp_path, nranks, 60, zlag, per_rank=64, devices='own', exchange='rccl')


@pytest.mark.parametrize('nranks', _ranks() or [2])
@pytest.mark.parametrize('zlag,shape', [
    (0, dict(sampler='demc')), (4, dict(sampler='demc')), (0, dict(sampler='snooker')),
    (0, dict(sampler='mrw', fx=1)), (1, dict(sampler='demc', thinning=2))],
    ids=lambda v: '-'.join(f'{k}{x}' for k, x in v.items()) if isinstance(v, dict) else f'zlag{v}')
def test_peer_to_peer_hand_off_across_gpus(tmp_path, nranks, zlag, shape):
    """The hand-laid-out loops' peer-to-peer instances (courier wave and all) with the ranks on
    different devices: IPC handles opened across devices, rows stored over xGMI at system
    scope, polled locally."""
    lean_ranks_case(tmp_path, nranks, 96, zlag, devices='own', **shape)


def test_config_c3_sharded_over_the_node(tmp_path):
    """BASELINE config C3 as BASELINE.json states it: 2048 chains over the node's GPUs."""
    nranks = max(_ranks())
    for zlag in (0, 4):
        d = tmp_path / f'zlag{zlag}'
        d.mkdir()
        lean_ranks_case(d, nranks, 64, zlag, per_rank=2048//nranks, devices='own')


@pytest.mark.parametrize('launcher', ['socket', 'torchrun'])
def test_sample_sharded_over_two_gpus(tmp_path, launcher):
    """mc3_amd.sample() end to end over two GPUs (defaults: the strict lockstep) == one rank
    running the checker."""
    p2p_case(tmp_path, 2, 'demc', 1, 300, launcher, env={'P2P_WORKER_DEVICES': 'own'})
