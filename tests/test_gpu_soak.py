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


# ---- the regime the bench line is measured in, at BASELINE.json's sizes -----------------------
# bench.py times C3-in-one-context, C4 and C5 in launches of thousands of refreshes; their
# kernels make up their work at run time (resident_async_kernel: blocks of sixteen chains from
# the proposals that are there; wavelet_pipe_kernel: slots of 21 turns over the chains of a
# workgroup; lean_demc_kernel BATCH: the draws of 64 refreshes at once), so a handful of
# refreshes does not reach the state the numbers come from.  These do, each split over launches
# (the second resumes what the first left), the kernel asserted by name, exact vs the oracle.
def _steady_state_soak(prob, launches, kernel, plan_checks):
    from mc3_amd import engine
    nthreads = min(64, len(os.sched_getaffinity(0)))
    with engine.DeviceChains(**prob) as gpu:
        for key, want in plan_checks.items():
            assert gpu.plan[key] == want, (key, gpu.plan)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        assert sum(launches) == gpu.nzchain
        gpu.run(launches[0], sync=False)          # the GPU works while the host cores do
        orc.run(nthreads=nthreads)
        gpu.sync()
        assert kernel in gpu.last_kernel, gpu.last_kernel
        for n in launches[1:]:
            gpu.run(n)
            assert kernel in gpu.last_kernel, gpu.last_kernel
        plan = gpu.plan
        for key, want in plan_checks.items():
            assert plan[key] == want, (key, plan)
        assert plan['recoveries'] == 0, 'a persistent launch gave up'
        Z, lp, zc = gpu.rows()
        st = gpu.state()
    assert np.array_equal(Z, orc.Z)
    assert np.array_equal(lp, orc.log_post)
    assert np.array_equal(zc, orc.zchain)
    assert np.array_equal(st['numaccept'], orc.numaccept)
    assert np.array_equal(st['outbounds'], orc.outbounds)
    assert np.array_equal(st['best_chisq'], orc.best_chisq)
    assert np.array_equal(st['best_iter'], orc.best_iter)
    assert np.array_equal(st['X'], orc.X)


def test_soak_c5_steady_state_of_the_async_resident_kernel():
    """C5 (512 chains x 1e5 points x 10 coefficients, priors, bounds): 640 refreshes through
    resident_async_kernel in launches of 500 + 140 -- long past the some hundred rounds its
    columns take to spread out, with the block that does not divide rotating between them."""
    nref = 640
    prob = H.linear_problem(ndata=100000, nchains=512, nsamples=512*nref)
    _steady_state_soak(prob, (500, 140), 'resident_async_kernel',
                       {'resident': 1, 'owner_waves': 3})


def test_soak_c4_steady_state_of_the_pipelined_wavelet_kernel():
    """C4 (256 chains x 2^14 points, wavelet likelihood, snooker): 2 100 iterations = 100 of
    pipe_chain_wave_one's batches of 21 turns, in launches of 1 500 + 600."""
    nref = 2100
    prob = H.wavelet_problem(ndata=16384, nchains=256, nsamples=256*nref)
    _steady_state_soak(prob, (1500, 600), 'wavelet_pipe_kernel', {'waves_per_chain': 8})


def test_soak_c3_in_one_context():
    """C3's ensemble on one GPU (2 048 chains x 1 000 points, demc): 5 120 iterations through
    lean_demc_kernel with the draws of 64 refreshes at once, two waves per SIMD, in launches
    of 3 000 + 2 120."""
    nref = 5120
    prob = H.quad_problem(ndata=1000, nchains=2048, nsamples=2048*nref, sampler='demc')
    _steady_state_soak(prob, (3000, 2120), 'lean_demc_kernel', {'fast': 2})
