"""Long runs of the headline kernels against the oracle, compared exactly (the pytest
form of tests/soak.py): 200 000 iterations x 256 chains of C2 = 51 M chain-steps through
lean_demc_kernel in 13 persistent launches, and a shorter snooker run through
lean_snooker_kernel.  The oracle side runs with OpenMP on the host cores."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('sampler,niter,fast', [('demc', 200000, 2), ('snooker', 40000, 2)])
def test_soak_c2(sampler, niter, fast):
    from mc3_amd import engine
    prob = H.quad_problem(ndata=1000, nchains=256, nsamples=256*niter, sampler=sampler)
    orc = H.make_oracle(prob)
    Z0, lp0 = H.initial_population(orc)
    orc.set_population(Z0, lp0)
    with engine.DeviceChains(**prob) as gpu:
        assert gpu.plan['fast'] == fast
        gpu.set_population(Z0, lp0)
        gpu.run(sync=False)                     # the GPU works while the host cores do
        orc.run(nthreads=min(64, len(os.sched_getaffinity(0))))
        gpu.sync()
        assert gpu.plan['fast'] == fast, 'a persistent launch gave up'
        Z, lp, zc = gpu.rows()
        st = gpu.state()
    assert np.array_equal(Z, orc.Z)
    assert np.array_equal(lp, orc.log_post)
    assert np.array_equal(zc, orc.zchain)
    assert np.array_equal(st['numaccept'], orc.numaccept)
    assert np.array_equal(st['outbounds'], orc.outbounds)
    assert np.array_equal(st['best_chisq'], orc.best_chisq)
    assert np.array_equal(st['best_iter'], orc.best_iter)


@pytest.mark.parametrize('sampler,niter,thinning', [('demc', 4000, 1), ('snooker', 3000, 1),
                                                    ('demc', 900, 3)])
def test_soak_resident_tiled_plan(sampler, niter, thinning):
    """Thousands of iterations of the one-launch tiled plan (64 chains x 30 000 points x 10
    coefficients: 15 tiles x 4 columns = 60 workgroups handing proposals and tile sums over
    inside the launch) against the oracle, compared exactly: a slot reused too early, a sum
    read before it was written or a row read across iterations shows up as a different
    chain, not as a crash."""
    from mc3_amd import engine
    nchains = 64
    prob = H.linear_problem(ndata=30000, nmodel=10, nchains=nchains,
                            nsamples=nchains*niter*thinning, sampler=sampler,
                            thinning=thinning)
    with engine.DeviceChains(**prob) as gpu:
        assert gpu.plan['resident'] == 1
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        gpu.run(sync=False)
        orc.run(nthreads=min(64, len(os.sched_getaffinity(0))))
        gpu.sync()
        assert gpu.plan['resident'] == 1 and gpu.plan['recoveries'] == 0, 'the launch gave up'
        Z, lp, zc = gpu.rows()
        st = gpu.state()
    assert np.array_equal(Z, orc.Z)
    assert np.array_equal(lp, orc.log_post)
    assert np.array_equal(zc, orc.zchain)
    assert np.array_equal(st['numaccept'], orc.numaccept)
    assert np.array_equal(st['outbounds'], orc.outbounds)
    assert np.array_equal(st['best_chisq'], orc.best_chisq)
    assert np.array_equal(st['best_iter'], orc.best_iter)
