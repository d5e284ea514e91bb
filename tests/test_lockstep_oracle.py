"""CPU tests of the lockstep oracle (the GPU's bit-exact checker): it must be
deterministic, independent of threading and of how chains are sharded, agree
with the reference-pinned kernels on chisq, and sample the same posterior as the
reference-pinned sequential oracle."""
import numpy as np
import pytest

from oracle import kernels as ok
from oracle import chain_sequential as cseq
import helpers as H


def run_oracle(prob, nthreads=1, **extra):
    orc = H.make_oracle(prob, **extra)
    Z0, lp0 = H.initial_population(orc)
    orc.set_population(Z0, lp0)
    orc.run(nthreads=nthreads)
    return orc


@pytest.mark.parametrize('builder', [H.quad_problem, H.shared_problem,
                                     H.wavelet_problem, H.linear_problem])
def test_lockstep_chisq_matches_reference_kernels(builder):
    prob = builder()
    orc = H.make_oracle(prob)
    rs = np.random.RandomState(0)
    for _ in range(5):
        p = orc.params + rs.normal(0, 1, len(orc.params))*np.abs(orc.pstep)
        for s, src in zip(orc.ishare, orc.share_src):
            p[s] = p[src]
        model = orc.eval_model(p)
        if prob.get('wlike'):
            want = ok.dwt_chisq(model, prob['data'], p, orc.prior, orc.priorlow,
                                orc.priorup)
        else:
            want = ok.chisq(model, prob['data'], prob['uncert'], p, orc.prior,
                            orc.priorlow, orc.priorup)
        got = orc.eval_chisq(p)[0]
        np.testing.assert_allclose(got, want, rtol=1e-12)
    # model values: Horner/dot with fma vs the plain NumPy expression
    if prob['model'] == 'poly':
        x = prob['indep']
        nm = orc.nmodel
        want = sum(p[k]*x**k for k in range(nm))
        np.testing.assert_allclose(orc.eval_model(p), want, rtol=1e-12, atol=1e-12)
    else:
        np.testing.assert_allclose(orc.eval_model(p), p[:orc.nmodel] @ prob['indep'],
                                   rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('sampler', ['demc', 'snooker', 'mrw'])
def test_lockstep_deterministic_and_thread_invariant(sampler):
    prob = H.quad_problem(sampler=sampler, fepsilon=0.01 if sampler == 'demc' else 0.0)
    a = run_oracle(prob, nthreads=1)
    b = run_oracle(prob, nthreads=4)
    np.testing.assert_array_equal(a.Z, b.Z)
    np.testing.assert_array_equal(a.log_post, b.log_post)
    np.testing.assert_array_equal(a.numaccept, b.numaccept)
    np.testing.assert_array_equal(a.outbounds, b.outbounds)
    assert np.all(a.zchain[a.M0:] == np.tile(np.arange(a.nchains), a.nzchain))
    assert a.numaccept.sum() > 0
    # another seed gives another trajectory
    c = run_oracle(dict(prob, seed=999))
    assert not np.array_equal(a.Z, c.Z)


@pytest.mark.parametrize('sampler,thinning', [('demc', 1), ('snooker', 1), ('demc', 3)])
def test_lockstep_sharding_invariance(sampler, thinning):
    """Two 'ranks' that each own half of the chains and exchange the new Z rows
    after every refresh reproduce the unsharded run bit for bit."""
    prob = H.quad_problem(sampler=sampler, thinning=thinning, nchains=8,
                          nsamples=8*60*thinning)
    full = run_oracle(prob)
    halves = [H.make_oracle(prob, chain0=0, nlocal=4),
              H.make_oracle(prob, chain0=4, nlocal=4)]
    Z0, lp0 = H.initial_population(halves[0])
    for h in halves:
        h.set_population(Z0, lp0)
    n = full.nchains
    for R in range(full.nzchain):
        for h in halves:
            h.run(nrefresh=1)
        r0 = full.M0 + R*n
        # the all-gather: every rank receives the other's block of rows
        halves[0].Z[r0 + 4:r0 + 8] = halves[1].Z[r0 + 4:r0 + 8]
        halves[1].Z[r0:r0 + 4] = halves[0].Z[r0:r0 + 4]
    np.testing.assert_array_equal(halves[0].Z, full.Z)
    np.testing.assert_array_equal(halves[1].Z, full.Z)
    acc = halves[0].numaccept + halves[1].numaccept
    np.testing.assert_array_equal(acc, full.numaccept)
    np.testing.assert_array_equal(halves[0].outbounds + halves[1].outbounds,
                                  full.outbounds)


def test_lockstep_samples_the_reference_posterior():
    """Same target distribution as the reference's sequential sampler: posterior
    means and standard deviations agree within Monte-Carlo error."""
    x, data, uncert = H.tutorial_data(100)
    common = dict(params=np.array([10.0, -2.0, 0.1]), pstep=np.array([1.0, 0.5, 0.1]),
                  pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]))
    seq = cseq.run_sequential(21, 22, data=data, uncert=uncert, func=H.quad,
                              indparams=[x], nchains=16, nsamples=4e4,
                              sampler='snooker', **common)
    for sampler in ('demc', 'snooker'):
        prob = H.quad_problem(ndata=100, nchains=16, nsamples=4e4, sampler=sampler)
        orc = run_oracle(prob)
        burn = orc.M0 + 16*300
        a = orc.Z[burn:]
        b = seq.Z[burn:]
        sd = b.std(axis=0)
        # effective sample size is >> 400 for both, so 0.15 sigma is generous
        assert np.all(np.abs(a.mean(axis=0) - b.mean(axis=0)) < 0.15*sd)
        assert np.all(np.abs(a.std(axis=0)/sd - 1.0) < 0.15)
        rate = orc.numaccept.sum()/(orc.nzchain*orc.nchains)
        assert 0.05 < rate < 0.6


# ---- the declared multi-GPU cadence (zlag >= 1, oracle_lockstep.c ls_run) -----------------
@pytest.mark.parametrize('sampler,thinning,zlag', [('demc', 1, 1), ('demc', 1, 4),
                                                   ('snooker', 1, 3), ('demc', 3, 2)])
def test_lagged_cadence_is_sharding_invariant_and_needs_only_late_rows(sampler, thinning, zlag):
    """Two 'ranks' that own half of the chains each reproduce the unsharded run of the same
    zlag bit for bit -- even though a rank receives the other's rows of refresh R only AFTER
    it has computed refresh R + zlag (the point of the cadence: a row has zlag refreshes' time
    to arrive; until then its place holds NaNs, which would poison a chain that looked)."""
    prob = H.quad_problem(sampler=sampler, thinning=thinning, nchains=8,
                          nsamples=8*50*thinning)
    full = run_oracle(prob, zlag=zlag)
    strict = run_oracle(prob)
    assert not np.array_equal(full.Z, strict.Z)          # (another cadence, another trajectory)
    halves = [H.make_oracle(prob, chain0=0, nlocal=4, zlag=zlag),
              H.make_oracle(prob, chain0=4, nlocal=4, zlag=zlag)]
    Z0, lp0 = H.initial_population(halves[0])
    for h in halves:
        h.set_population(Z0, lp0)
        h.Z[h.M0:] = np.nan
    n = full.nchains
    for R in range(full.nzchain):
        for h in halves:
            h.run(nrefresh=1)
        # the exchange delivers what was produced zlag refreshes ago
        Rl = R - zlag
        if Rl >= 0:
            r0 = full.M0 + Rl*n
            halves[0].Z[r0 + 4:r0 + 8] = halves[1].Z[r0 + 4:r0 + 8]
            halves[1].Z[r0:r0 + 4] = halves[0].Z[r0:r0 + 4]
    for R in range(max(0, full.nzchain - zlag), full.nzchain):
        r0 = full.M0 + R*n
        halves[0].Z[r0 + 4:r0 + 8] = halves[1].Z[r0 + 4:r0 + 8]
        halves[1].Z[r0:r0 + 4] = halves[0].Z[r0:r0 + 4]
    np.testing.assert_array_equal(halves[0].Z, full.Z)
    np.testing.assert_array_equal(halves[1].Z, full.Z)
    np.testing.assert_array_equal(halves[0].numaccept + halves[1].numaccept, full.numaccept)


def test_lagged_cadence_first_refreshes_see_the_initial_snapshot():
    """DEMC with zlag = 2: refreshes 0..2 all draw their partners from the first nchains rows
    of the initial population (there is no refresh R - 3 yet), refresh 3 from refresh 0 --
    checked by poisoning exactly the rows that must not be looked at."""
    prob = H.quad_problem(sampler='demc', nchains=8, nsamples=8*6)
    ref = run_oracle(prob, zlag=2)
    orc = H.make_oracle(prob, zlag=2)
    Z0, lp0 = H.initial_population(orc)
    orc.set_population(Z0, lp0)
    for R in range(6):
        orc.run(nrefresh=1)
        # hide the rows just written from everybody: own state lives in X, and no refresh
        # before R + 3 may look at them
        keep = orc.Z[orc.M0 + R*8:orc.M0 + (R + 1)*8].copy()
        orc.Z[orc.M0 + R*8:orc.M0 + (R + 1)*8] = np.nan
        if R >= 2:
            back = R - 2
            orc.Z[orc.M0 + back*8:orc.M0 + (back + 1)*8] = ref.Z[ref.M0 + back*8:
                                                                 ref.M0 + (back + 1)*8]
        assert np.array_equal(keep, ref.Z[ref.M0 + R*8:ref.M0 + (R + 1)*8])


def test_lagged_cadence_samples_the_same_posterior():
    """zlag changes which past states a chain looks at, not what it samples: means and widths
    of a zlag = 4 run agree with the strict run's within Monte-Carlo error (256 chains would
    be the production shape; 16 keeps the test short and makes staleness matter more)."""
    for sampler in ('demc', 'snooker'):
        prob = H.quad_problem(ndata=100, nchains=16, nsamples=4e4, sampler=sampler)
        a = run_oracle(prob)
        b = run_oracle(prob, zlag=4)
        burn = a.M0 + 16*300
        sd = a.Z[burn:].std(axis=0)
        assert np.all(np.abs(a.Z[burn:].mean(axis=0) - b.Z[burn:].mean(axis=0)) < 0.15*sd)
        assert np.all(np.abs(b.Z[burn:].std(axis=0)/sd - 1.0) < 0.15)
        ra = a.numaccept.sum()/(a.nzchain*a.nchains)
        rb = b.numaccept.sum()/(b.nzchain*b.nchains)
        assert abs(ra - rb) < 0.05


@pytest.mark.parametrize('sampler,thinning', [('demc', 1), ('snooker', 2), ('mrw', 1)])
def test_worker_processes_equal_threads(sampler, thinning):
    """oracle_lockstep.c ls_run_worker (bench.py's cpu_baseline layout "processes", SURVEY.md
    8d(ii)): the run laid out as the reference's fan-out -- worker PROCESSES, worker i advancing
    chains i, i + ncpu, ... (mc3/chain.py:163,190-191), one meeting per refresh in shared
    memory -- equals the threaded run bit for bit, whatever the number of workers, also when
    they do not divide the chains and across two calls."""
    prob = H.quad_problem(nchains=13, nsamples=13*120*thinning, sampler=sampler,
                          thinning=thinning)
    ref = H.make_oracle(prob)
    Z0, lp0 = H.initial_population(ref)
    ref.set_population(Z0, lp0)
    ref.run(nthreads=3)
    for first, second in ((4, 3), (1, 13), (16, 2)):
        orc = H.make_oracle(prob, shared=True)
        orc.set_population(Z0, lp0)
        assert orc.run_processes(50, first) > 0.0
        orc.run_processes(orc.nzchain - 50, second)
        for key in ('Z', 'log_post', 'zchain', 'numaccept', 'outbounds', 'X', 'chisq',
                    'best_chisq', 'best_iter', 'bestx'):
            np.testing.assert_array_equal(getattr(orc, key), getattr(ref, key), err_msg=key)
    with pytest.raises(RuntimeError, match='shared=True'):
        H.make_oracle(prob).run_processes(1, 2)


def test_worker_processes_that_fail_end_the_run(monkeypatch):
    """A worker that never reaches the first meeting (test hook), or arguments the workers
    would refuse: the parent neither spins for ever nor leaves workers behind -- it calls the
    meetings off, ends the others and raises."""
    import time
    prob = H.quad_problem(nchains=8, nsamples=8*40, sampler='demc')
    orc = H.make_oracle(prob, shared=True)
    Z0, lp0 = H.initial_population(orc)
    orc.set_population(Z0, lp0)
    with pytest.raises(ValueError, match='no room'):
        orc.run_processes(orc.nzchain + 1, 2)
    with pytest.raises(ValueError, match='ncpu'):
        orc.run_processes(1, 0)
    monkeypatch.setenv('ORACLE_TEST_WORKER_DIES', '1')
    t0 = time.time()
    with pytest.raises(RuntimeError, match='worker processes failed'):
        orc.run_processes(10, 3)
    assert time.time() - t0 < 30.0
    monkeypatch.delenv('ORACLE_TEST_WORKER_DIES')
    assert orc.run_processes(10, 3) > 0.0          # and the run can go on afterwards
