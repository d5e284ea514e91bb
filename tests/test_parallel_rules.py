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

    loads = None                      # (device, grid, capacity) of every rank, when it matters

    def all_gather(self, obj):
        if isinstance(obj, tuple) and len(obj) == 3:          # the co-residency vote
            out = list(self.loads) if self.loads else \
                [(a[0] if isinstance(a, tuple) else a, 0, 0) for a in self.answers]
        else:
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
    """The peer-to-peer instances of the headline loops run a second wave per chain (the
    courier); the two-wave workgroups of more than two ranks of 256 chains are not all
    co-resident on ONE device, so such a set-up runs without it -- on every rank alike,
    whichever device it sits on.  Decided BEFORE the engine exists (engine_options: the
    library sizes its persistent grid from the very kernel instance the flag selects; no
    environment variable is touched), and checked against the engine at set-up."""
    class DeviceComm(FixedComm):
        def __init__(self, rank, devices):
            super().__init__(rank, devices)

        def broadcast(self, obj):
            return obj

    opts = DeviceComm(0, [0, 1]).engine_options(device=0)
    assert opts == {'zlag': 0, 'p2p': True, 'no_courier': False}     # strict by default
    opts = DeviceComm(0, [0, 1]).engine_options(device=0, zlag='auto')
    assert opts == {'zlag': parallel.CADENCE_ZLAG, 'p2p': True, 'no_courier': False}
    assert DeviceComm(1, [0, 0]).engine_options(device=0)['no_courier'] is False
    assert DeviceComm(3, [0, 0, 0, 1]).engine_options(device=1)['no_courier'] is True
    assert DeviceComm(0, [0, 1]).engine_options(device=0, thinning=3, zlag='auto')['zlag'] == 2
    os.environ['MC3HIP_ZLAG'] = '3'
    try:
        assert DeviceComm(0, [0, 1]).engine_options(device=0)['zlag'] == 3
        assert parallel.SingleRank().engine_options() == {'zlag': 3}
        os.environ['MC3HIP_ZLAG'] = 'auto'
        assert DeviceComm(0, [0, 1]).engine_options(device=0, thinning=2)['zlag'] == 2
    finally:
        del os.environ['MC3HIP_ZLAG']
    with pytest.raises(ValueError, match='zlag'):
        parallel.resolve_zlag(-1)
    assert DeviceComm(0, [0, 1]).engine_options(device=0, zlag=0)['zlag'] == 0
    assert 'MC3HIP_NO_COURIER' not in os.environ
    # an engine created otherwise is refused at set-up, on every rank
    crowded = [(0, False), (0, False), (0, False), (1, False)]
    eng = engine(1)
    eng.no_courier = False
    with pytest.raises(RuntimeError, match='engine_options'):
        FixedComm(3, crowded)._check_shared_device(eng)
    eng.no_courier = True
    FixedComm(3, crowded)._check_shared_device(eng)
    # one rank decides nothing
    assert parallel.SingleRank().engine_options() == {}
    assert parallel.SingleRank().engine_options(zlag=3) == {'zlag': 3}


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


def test_ranks_that_share_a_device_must_fit_it_together(clean_env):
    """Every library checks its own persistent grid against the device; the SUM over the ranks
    on one device is the communicator's business (their kernels wait for each other's rows)."""
    eng = types.SimpleNamespace(_device=0, nlocal=256, no_courier=True,
                                plan={'resident': 0, 'persistent': 16384, 'capacity': 1024,
                                      'chains_per_workgroup': 1})
    comm = FixedComm(0, [(0, False)]*5)
    comm.loads = [(0, 256, 1024)]*5                     # 1280 workgroups, 1024 fit
    with pytest.raises(RuntimeError, match='co-resident'):
        comm._check_shared_device(eng)
    comm.loads = [(0, 256, 2048)]*5
    eng.plan['capacity'] = 2048
    comm._check_shared_device(eng)                      # 1280 of 2048: fine
    comm = FixedComm(0, [(0, False), (1, False), (2, False)])
    comm.loads = [(0, 256, 256), (1, 256, 256), (2, 256, 256)]
    eng.plan['capacity'] = 256
    comm._check_shared_device(eng)                      # a GPU each: nothing to add up
