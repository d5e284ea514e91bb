"""The cross-GPU peer-to-peer hand-off, exercised with two PROCESSES on one GPU:
each rank owns half of the chains and a full Z replica, opens the other's replica
through HIP IPC, publishes its rows into both and polls its own.  Must reproduce
the single-rank run bit for bit (draws are keyed by global chain id)."""
import os

import numpy as np
import pytest

import helpers as H
from ranks import launch_ranks

# (the torchrun case: the first `import torch` on a fresh box can take minutes per worker)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1200)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('nproc,sampler,thinning,niter,launcher', [
    (2, 'demc', 1, 400, 'socket'), (2, 'snooker', 2, 200, 'socket'),
    (4, 'demc', 1, 300, 'socket'), (4, 'snooker', 1, 300, 'socket'),
    (8, 'demc', 1, 200, 'socket'), (2, 'demc', 1, 300, 'torchrun'),
    # (more ranks than replica pointers are kept in registers: handoff.h PeerTable's slow way)
    (9, 'demc', 1, 120, 'socket'), (9, 'snooker', 1, 120, 'socket')])
def test_processes_peer_to_peer(tmp_path, nproc, sampler, thinning, niter, launcher):
    """The STRICT lockstep across ranks (zlag = 0: every iteration waits for the rows of the
    one before, wherever they were made)."""
    p2p_case(tmp_path, nproc, sampler, thinning, niter, launcher, zlag=0)


@pytest.mark.parametrize('nproc,sampler,thinning,zlag,delay_ns', [
    (2, 'demc', 1, None, 0), (2, 'demc', 1, None, 3000), (4, 'demc', 1, None, 2000),
    (8, 'demc', 1, None, 2000), (2, 'demc', 3, None, 2000), (8, 'demc', 2, 1, 1000),
    (2, 'snooker', 1, None, 2000), (8, 'snooker', 1, 2, 1000), (2, 'mrw', 1, None, 2000),
    (2, 'demc', 1, 1, 5000), (9, 'demc', 1, None, 1000)])
def test_processes_in_the_declared_cadence(tmp_path, nproc, sampler, thinning, zlag, delay_ns):
    """mc3_amd.sample() over several ranks in the DECLARED MULTI-GPU CADENCE (opt-in:
    zlag='auto' = parallel.CADENCE_ZLAG refreshes' worth; here also 1 and 2), with the
    rows for the other ranks' replicas held back 1-5 us (MC3HIP_TEST_REMOTE_DELAY_NS: ranks on
    one GPU then see each other as if an xGMI hop apart) -- identical, bit for bit, to ONE
    rank running the oracle's restatement of the same cadence.  Rows that arrive later than
    zlag iterations after they were made (5 us against zlag = 1) are waited for: slower, the
    same results."""
    from mc3_amd import parallel
    env = {'MC3HIP_TEST_REMOTE_DELAY_NS': str(delay_ns)} if delay_ns else {}
    if zlag is None:
        # (zlag='auto': CADENCE_ZLAG iterations' worth of refreshes; ONE rank must be asked for
        # the same number -- the strict lockstep is the default of every world size)
        want = -(-parallel.CADENCE_ZLAG//thinning)
        p2p_case(tmp_path, nproc, sampler, thinning, 300, 'socket', env=env, zlag='auto',
                 want_zlag=want)
    else:
        p2p_case(tmp_path, nproc, sampler, thinning, 300, 'socket', env=env, zlag=zlag)


def test_peer_to_peer_without_the_courier(tmp_path):
    """The headline loop's peer-to-peer instance with the chain wave storing into the other
    ranks' replicas itself (MC3HIP_NO_COURIER=1; by default a second wave of the workgroup
    carries the rows, lean_demc_kernel COURIER -- the 2-rank demc cases above and the abort
    case below; more than two ranks on ONE device run without it, mc3_amd/parallel.py)."""
    p2p_case(tmp_path, 2, 'demc', 1, 400, 'socket', env={'MC3HIP_NO_COURIER': '1'}, zlag=0)


def p2p_case(tmp_path, nproc, sampler, thinning, niter, launcher, env=None, zlag=None,
             want_zlag=None):
    import mc3_amd
    from mc3_amd import utils as mu
    from oracle_engine import OracleChains
    out_path = str(tmp_path / 'p2p')
    nchains = 16 if 16 % nproc == 0 else 2*nproc          # (equal blocks of chains per rank)
    more = [f'zlag={zlag}'] if zlag is not None else []
    launch_ranks(nproc, 'p2p_worker.py', out_path, sampler, thinning, niter,
                 f'nchains={nchains}', *more, launcher=launcher, env=env)

    x, data, uncert = H.tutorial_data(100)
    np.random.seed(100)
    single = mc3_amd.sample(
        data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
        indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
        pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
        sampler=sampler, nsamples=nchains*niter*thinning, burnin=10, nchains=nchains,
        thinning=thinning, grtest=True, log=mu.Log(verb=0),
        engine_factory=OracleChains, seed=99, zlag=zlag if want_zlag is None else want_zlag)
    for rank in range(nproc):
        got = np.load(f'{out_path}.rank{rank}.npz')
        for key in ('posterior', 'zchain', 'log_post', 'chisq', 'bestp',
                    'acceptance_rate', 'best_log_post', 'medianp'):
            np.testing.assert_array_equal(got[key], single[key], err_msg=key)
        np.testing.assert_allclose(got['stdp'], single['stdp'], rtol=1e-12)
        assert int(got['zlag']) == int(single['zlag'])      # (the cadence is on record)


def launch(nproc, script, *args, env=None, timeout=1000):
    return launch_ranks(nproc, script, *args, env=env, timeout=timeout)


def lean_ranks_case(tmp_path, nranks, nref, zlag, delay_ns=0, **shape):
    """`nranks` processes on the one GPU, each running the hand-laid-out loop of its shape over
    its block of chains (tests/c3_worker.py asserts plan.fast == 2): every replica must equal
    the oracle's run of the whole ensemble in the same cadence."""
    per_rank = shape.get('per_rank', 256)
    nchains = per_rank*nranks
    prob = H.lean_shape_problem(nchains, nref, **{k: v for k, v in shape.items()
                                                  if k not in ('per_rank', 'turns', 'devices',
                                                               'exchange')})
    env = {'MC3HIP_TEST_REMOTE_DELAY_NS': str(delay_ns)} if delay_ns else {}
    if 'turns' in shape:       # (several chains per wave: pretend that fewer workgroups fit)
        env['MC3HIP_PERSIST_CAPACITY'] = str(-(-per_rank//shape['turns']))
    if shape.get('exchange') == 'rccl' and shape.get('devices') != 'own':
        # (the ranks share ONE GPU: each is told it is a host of its own, tests/c3_worker.py)
        env['MC3_TEST_RCCL_RANKS_AS_HOSTS'] = '1'
        env.setdefault('NCCL_DEBUG', 'ERROR')
    orc = H.make_oracle(prob, zlag=zlag)
    Z0, lp0 = H.initial_population(orc)
    out_path = str(tmp_path / 'c3')
    np.savez(out_path + '.pop.npz', Z0=Z0, lp0=lp0)
    orc.set_population(Z0, lp0)
    orc.run(nthreads=min(32, len(os.sched_getaffinity(0))))
    launch(nranks, 'c3_worker.py', out_path, nref, f'zlag={zlag}',
           *[f'{k}={v}' for k, v in shape.items()], env=env or None)
    for rank in range(nranks):
        got = np.load(f'{out_path}.rank{rank}.npz')
        s = slice(per_rank*rank, per_rank*(rank + 1))
        assert int(got['zlag']) == zlag
        np.testing.assert_array_equal(got['Z'], orc.Z, err_msg=f'rank {rank}')
        np.testing.assert_array_equal(got['lp'], orc.log_post)
        np.testing.assert_array_equal(got['zc'], orc.zchain)
        np.testing.assert_array_equal(got['numaccept'], orc.numaccept[s])
        np.testing.assert_array_equal(got['X'], orc.X[s])
        np.testing.assert_array_equal(got['chisq'], orc.chisq[s])
        np.testing.assert_array_equal(got['best_chisq'], orc.best_chisq[s])
    total_out = sum(np.load(f'{out_path}.rank{r}.npz')['outbounds'] for r in range(nranks))
    np.testing.assert_array_equal(total_out, orc.outbounds)


@pytest.mark.parametrize('zlag,delay_ns', [(0, 0), (4, 0), (4, 2000), (1, 3000)])
def test_config_c3_eight_ranks_of_256_chains(tmp_path, zlag, delay_ns):
    """BASELINE config C3 at its real shape: 2048 chains x ndata 1000, demc, sharded over
    8 ranks of 256 chains through the peer-to-peer hand-off (here: eight processes on the
    one GPU) -- in the strict lockstep and in the declared cadence, the latter also with
    every row for another rank's replica held back 2-3 us."""
    lean_ranks_case(tmp_path, 8, 48, zlag, delay_ns)


@pytest.mark.parametrize('nranks', [2, 8])
@pytest.mark.parametrize('shape', [
    dict(sampler='demc', fx=1), dict(sampler='demc', thinning=3), dict(sampler='demc', fx=1,
                                                                        thinning=2),
    dict(sampler='snooker'), dict(sampler='snooker', fx=1), dict(sampler='snooker', thinning=3),
    dict(sampler='mrw'), dict(sampler='mrw', fx=1), dict(sampler='mrw', thinning=2)],
    ids=lambda d: '-'.join(f'{k}{v}' for k, v in d.items()))
def test_hand_laid_out_loops_peer_to_peer(tmp_path, nranks, shape):
    """Every hand-laid-out loop in its peer-to-peer flavour -- fixed / shared coefficients and
    priors, thinning > 1, snooker, random walk (mc3/chain.py:195-220) -- with 2 ranks (courier
    wave) and 8 ranks (the chain wave stores itself) on the one GPU, rows held back 1.5 us,
    in the declared cadence (demc, snooker) / as ever (mrw looks at nobody)."""
    lean_ranks_case(tmp_path, nranks, 40, 2, 1500, per_rank=24, ndata=300, **shape)


@pytest.mark.parametrize('nranks,turns', [(2, 3), (4, 2), (2, 4)])
@pytest.mark.parametrize('shape,zlag', [
    (dict(sampler='demc'), 2), (dict(sampler='demc'), 0), (dict(sampler='demc', fx=1), 1),
    (dict(sampler='snooker'), 2), (dict(sampler='snooker', fx=1), 0), (dict(sampler='mrw'), 4)],
    ids=lambda d: '-'.join(f'{k}{v}' for k, v in d.items()) if isinstance(d, dict) else f'zlag{d}')
def test_rows_loops_peer_to_peer(tmp_path, nranks, turns, shape, zlag):
    """kernels_rows.h peer to peer: every rank advances its chains two to four per wave, side by
    side (MC3HIP_PERSIST_CAPACITY pretends that fewer workgroups fit), and stores the rows of a
    wave's chains into every replica itself -- strict lockstep and the declared cadence, rows
    held back 1.5 us, ragged last waves (23 chains per rank)."""
    lean_ranks_case(tmp_path, nranks, 40, zlag, 1500, per_rank=23, ndata=300, turns=turns,
                    **shape)


@pytest.mark.parametrize('nranks,per_rank,ndata,nmodel,sampler,zlag', [
    (2, 48, 20000, 7, 'demc', 0), (4, 32, 30000, 10, 'snooker', 0),
    (4, 128, 100000, 10, 'demc', 0), (2, 48, 40000, 3, 'demc', 0),
    (4, 32, 40000, 3, 'snooker', 0), (4, 128, 100000, 10, 'demc', 2),
    (2, 48, 20000, 7, 'snooker', 1)])
def test_resident_tiled_plan_sharded_over_ranks(tmp_path, nranks, per_rank, ndata, nmodel,
                                                sampler, zlag):
    """The tiled plan in one launch per rank (resident_tile_kernel<KB, P2P>), the ensemble
    sharded over ranks through the peer-to-peer hand-off -- the last case is BASELINE config
    C5's own shape: 512 chains x 1e5 points x 10 coefficients over 4 ranks of 128 chains
    (here: four processes on the one GPU, which they must share: one column of chains per
    rank, 4 x 49 tiles = 196 workgroups).  Every rank's replica must equal the oracle's run of
    the whole ensemble."""
    nref = 12
    nchains = nranks*per_rank
    if nmodel == 3:         # the polynomial flavour (resident_poly_kernel<3, P2P>)
        prob = H.quad_problem(ndata=ndata, nchains=nchains, nsamples=nchains*nref,
                              sampler=sampler)
        plan = {'tile': 4096, 'vlanes': 1024, 'tile_tree': 0}
    else:
        prob = H.linear_problem(ndata=ndata, nmodel=nmodel, nchains=nchains,
                                nsamples=nchains*nref, sampler=sampler)
        plan = {'tile': 2048, 'vlanes': 1024, 'tile_tree': 2}
    orc = H.make_oracle(prob, plan=plan, zlag=zlag)
    Z0, lp0 = H.initial_population(orc)
    out_path = str(tmp_path / 'c5')
    np.savez(out_path + '.pop.npz', Z0=Z0, lp0=lp0)
    orc.set_population(Z0, lp0)
    orc.run(nthreads=min(32, len(os.sched_getaffinity(0))))
    launch(nranks, 'c5_worker.py', out_path, nref, per_rank, ndata, nmodel, sampler,
           f'zlag={zlag}', env={'MC3HIP_RESIDENT_MAXCOL': '1',
                                **({'MC3HIP_TEST_REMOTE_DELAY_NS': '2000'} if zlag else {})})
    for rank in range(nranks):
        got = np.load(f'{out_path}.rank{rank}.npz')
        s = slice(per_rank*rank, per_rank*(rank + 1))
        np.testing.assert_array_equal(got['Z'], orc.Z, err_msg=f'rank {rank}')
        np.testing.assert_array_equal(got['lp'], orc.log_post)
        np.testing.assert_array_equal(got['zc'], orc.zchain)
        np.testing.assert_array_equal(got['numaccept'], orc.numaccept[s])
        np.testing.assert_array_equal(got['X'], orc.X[s])
        np.testing.assert_array_equal(got['chisq'], orc.chisq[s])
        np.testing.assert_array_equal(got['best_chisq'], orc.best_chisq[s])
    total_out = sum(np.load(f'{out_path}.rank{r}.npz')['outbounds'] for r in range(nranks))
    np.testing.assert_array_equal(total_out, orc.outbounds)


@pytest.mark.parametrize('sampler', ['demc', 'snooker'])
def test_peer_to_peer_abort_is_recovered_by_all_ranks(tmp_path, sampler):
    """A persistent launch gives up on rank 0 in the middle of a two-rank run (test
    hook; in the field: a peer that died or a grid that was not resident).  Rank 1's
    waits time out, the ranks vote at the next report point, EVERY rank rolls back to
    its checkpoint and the lost refreshes are redone through the host-staged exchange:
    same output dictionary as the undisturbed single-rank run."""
    import mc3_amd
    from mc3_amd import utils as mu
    from oracle_engine import OracleChains
    niter = 200
    out_path = str(tmp_path / 'abort')
    res = launch(2, 'p2p_worker.py', out_path, sampler, 1, niter, 'abort=77', 'zlag=0')
    assert 'recovered' in res, res[-2000:]
    x, data, uncert = H.tutorial_data(100)
    np.random.seed(100)
    single = mc3_amd.sample(
        data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
        indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
        pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
        sampler=sampler, nsamples=16*niter, burnin=10, nchains=16,
        thinning=1, grtest=True, log=mu.Log(verb=0),
        engine_factory=OracleChains, seed=99)
    for rank in range(2):
        got = np.load(f'{out_path}.rank{rank}.npz')
        for key in ('posterior', 'zchain', 'log_post', 'chisq', 'bestp',
                    'acceptance_rate', 'best_log_post', 'medianp'):
            np.testing.assert_array_equal(got[key], single[key], err_msg=key)
        np.testing.assert_allclose(got['stdp'], single['stdp'], rtol=1e-12)


def test_resident_tiled_plan_abort_is_recovered_by_all_ranks(tmp_path):
    """The same vote / roll-back / redo when the kernel that gives up is the resident tiled
    plan's (two ranks of 32 chains, 10 tiles; rank 0's first chain gives up at refresh 7 of
    the second launch): the lost refreshes are redone through the host-staged exchange with
    the three-launch plan, and every replica still equals the oracle's run."""
    nref, per_rank, ndata, nmodel = 18, 32, 20000, 7
    nchains = 2*per_rank
    prob = H.linear_problem(ndata=ndata, nmodel=nmodel, nchains=nchains,
                            nsamples=nchains*nref, sampler='demc')
    orc = H.make_oracle(prob, plan={'tile': 2048, 'vlanes': 1024, 'tile_tree': 2})
    Z0, lp0 = H.initial_population(orc)
    out_path = str(tmp_path / 'c5abort')
    np.savez(out_path + '.pop.npz', Z0=Z0, lp0=lp0)
    orc.set_population(Z0, lp0)
    orc.run(nthreads=8)
    res = launch(2, 'c5_worker.py', out_path, nref, per_rank, ndata, nmodel, 'demc',
                 'abort=13', 'zlag=0', env={'MC3HIP_RESIDENT_MAXCOL': '1'})
    assert 'recovered' in res, res[-2000:]
    for rank in range(2):
        got = np.load(f'{out_path}.rank{rank}.npz')
        s = slice(per_rank*rank, per_rank*(rank + 1))
        np.testing.assert_array_equal(got['Z'], orc.Z, err_msg=f'rank {rank}')
        np.testing.assert_array_equal(got['lp'], orc.log_post)
        np.testing.assert_array_equal(got['numaccept'], orc.numaccept[s])
        np.testing.assert_array_equal(got['chisq'], orc.chisq[s])


@pytest.mark.skipif(not os.environ.get('MC3_TEST_RCCL_ON_ONE_GPU'),
                    reason='opt-in (MC3_TEST_RCCL_ON_ONE_GPU=1): RCCL over its socket transport '
                           'between ranks that share ONE GPU -- passed on three boxes in round 6 '
                           '(profiles/r06_rccl_ranks_on_one_gpu.txt), but a heavier run of the '
                           'same set-up (bench.py, 10 000 refreshes) coincided with the loss of '
                           'a box, so the suite does not run it unasked')
@pytest.mark.parametrize('nranks,zlag', [(2, 0), (2, 2), (4, 0), (4, 1)])
def test_rccl_all_gather_with_more_than_one_rank_on_the_one_gpu(tmp_path, nranks, zlag):
    """The collective of the north star with N > 1 ranks, on the box the suite runs on: one
    launch per refresh + ONE in-place ncclAllGather of the refresh's Z rows, replayed from a
    hipGraph, over 2 and 4 ranks.  RCCL refuses two ranks of one host on one device, so every
    rank is told it is a host of its own (NCCL_HOSTID) and RCCL connects them as the nodes of a
    cluster, through its socket transport over the loopback interface: not a fast path -- but
    the bootstrap, the ring, the all-gather in place, the captured graph and its teardown are
    RCCL's own with more than one rank (round 6: the first such run found mc3hip_destroy
    waiting for ever in ncclCommDestroy -- the captured graph must go first).  tests/c3_worker.py
    asserts RCCL's own rank count and plan.comm_graph > 0; every replica == the oracle's run of
    the whole ensemble, strict and in a cadence.  (With a GPU per rank: test_gpu_multidevice.py.)
    Reference: mc3/mcmc_driver.py:204-221,300-307."""
    lean_ranks_case(tmp_path, nranks, 45, zlag, per_rank=32, ndata=300, exchange='rccl')
