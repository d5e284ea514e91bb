"""Host-side rules of the multi-rank set-up that need no GPU: what Communicator decides when
several ranks share one device (a test set-up; a node gives every rank a GPU of its own)."""
import os
import types

import pytest

from mc3_amd import parallel


class FixedComm(parallel.Communicator):
    """A communicator whose all_gather answers with a prepared list (what the other ranks
    would have sent)."""

    def __init__(self, rank, answers):
        super().__init__(rank, len(answers))
        self.answers = answers

    def all_gather(self, obj):
        out = list(self.answers)
        out[self.rank] = obj
        return out


def engine(device, resident=False):
    return types.SimpleNamespace(_device=device, plan={'resident': int(resident)})


@pytest.fixture
def clean_env(monkeypatch):
    monkeypatch.delenv('MC3HIP_NO_COURIER', raising=False)
    monkeypatch.delenv('MC3HIP_RESIDENT_MAXCOL', raising=False)
    yield
    os.environ.pop('MC3HIP_NO_COURIER', None)


def test_courier_only_with_at_most_two_ranks_per_device(clean_env):
    """The peer-to-peer instance of the headline loop runs a second wave per chain (the
    courier, lean_demc_kernel COURIER); the two-wave workgroups of more than two ranks of 256
    chains are not all co-resident on ONE device, so such a set-up switches it off -- on every
    rank alike, whichever device it sits on."""
    FixedComm(0, [(0, False), (1, False)])._check_shared_device(engine(0))
    assert 'MC3HIP_NO_COURIER' not in os.environ          # a GPU each
    FixedComm(1, [(0, False), (0, False)])._check_shared_device(engine(0))
    assert 'MC3HIP_NO_COURIER' not in os.environ          # two on one device: still fits
    FixedComm(3, [(0, False), (0, False), (0, False), (1, False)])._check_shared_device(engine(1))
    assert os.environ.get('MC3HIP_NO_COURIER') == '1'     # three share device 0: all without


def test_shared_device_with_the_one_launch_tiled_plan_needs_a_share(clean_env, monkeypatch):
    """Ranks that share a device with the one-launch tiled plan must be told how they share
    it (MC3HIP_RESIDENT_MAXCOL): refused otherwise, on every rank."""
    answers = [(0, True), (0, True)]
    with pytest.raises(RuntimeError, match='MC3HIP_RESIDENT_MAXCOL'):
        FixedComm(0, answers)._check_shared_device(engine(0, resident=True))
    monkeypatch.setenv('MC3HIP_RESIDENT_MAXCOL', '2')
    FixedComm(0, answers)._check_shared_device(engine(0, resident=True))
    # a GPU each: nothing to say
    monkeypatch.delenv('MC3HIP_RESIDENT_MAXCOL')
    FixedComm(0, [(0, True), (1, True)])._check_shared_device(engine(0, resident=True))
