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
