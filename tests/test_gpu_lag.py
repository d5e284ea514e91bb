"""The declared multi-GPU cadence (mc3hip_problem.zlag >= 1) on ONE GPU: every kernel family
against the oracle's restatement of the same cadence (oracle_lockstep.c ls_run), bit for bit.

During refresh R a chain sees the others through the rows of refresh R - 1 - zlag (DEMC,
mc3/chain.py:221-232) / the history up to that refresh (snooker, chain.py:197-205); the first
nchains rows of the initial population stand in until there is such a refresh.  Results depend
on zlag and on nothing else -- not on the launch structure, not on how the chains are sharded
(tests/test_gpu_p2p.py runs the same cadence over several ranks).
"""
import numpy as np
import pytest

import helpers as H
from test_gpu_parity import assert_same, run_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from mc3_amd import engine
    return engine


# ---- the general chain (run_chain): every sampler, likelihood, model family, launch structure
@pytest.mark.parametrize('sampler', ['demc', 'snooker', 'mrw'])
@pytest.mark.parametrize('zlag', [1, 3])
def test_general_chain_in_the_lagged_cadence(eng, sampler, zlag):
    prob = H.quad_problem(nchains=9, nsamples=9*120, sampler=sampler, zlag=zlag,
                          fepsilon=0.01 if sampler == 'demc' else 0.0)
    orc, gpu = run_pair(eng, prob)                 # (traced: the general chain)
    assert gpu.plan['zlag'] == zlag and gpu.plan['fast'] == 0
    assert_same(orc, gpu)
    # ... and it is another trajectory than the strict lockstep's (mrw looks at nobody)
    strict = H.make_oracle(dict(prob, zlag=0), plan=gpu.plan)
    Z0, lp0 = H.initial_population(strict)
    strict.set_population(Z0, lp0)
    strict.run(nthreads=4)
    assert np.array_equal(strict.Z, orc.Z) == (sampler == 'mrw')
    gpu.close()


@pytest.mark.parametrize('kind', ['shared-snooker', 'wavelet', 'wavelet-8192', 'linear',
                                  'tiled-3launch', 'resident', 'per-refresh', 'thin'])
def test_every_kernel_family_in_the_lagged_cadence(eng, monkeypatch, kind):
    kw = {}
    trace = False
    if kind == 'shared-snooker':
        prob = H.shared_problem(zlag=2)
    elif kind == 'wavelet':
        prob = H.wavelet_problem(zlag=2)
    elif kind == 'wavelet-8192':                    # (nine-wave kernel, general chain wave)
        prob = H.wavelet_problem(ndata=8192, nchains=8, nsamples=8*30, zlag=2)
    elif kind == 'linear':
        prob = H.linear_problem(zlag=3)
    elif kind == 'tiled-3launch':
        monkeypatch.setenv('MC3HIP_NO_RESIDENT', '1')
        prob = H.linear_problem(ndata=20000, nmodel=7, nchains=16, nsamples=16*14, zlag=2)
    elif kind == 'resident':                        # (the hand-structured owner follows zlag)
        prob = H.linear_problem(ndata=20000, nmodel=7, nchains=16, nsamples=16*40, zlag=2)
    elif kind == 'per-refresh':
        prob = H.quad_problem(nchains=8, nsamples=8*60, sampler='demc', zlag=2)
        kw = dict(no_persist=True)
    else:
        prob = H.quad_problem(nchains=8, nsamples=8*40*3, sampler='snooker', thinning=3, zlag=2)
        trace = True
    orc, gpu = run_pair(eng, prob, trace=trace, **kw)
    plan = gpu.plan
    assert plan['zlag'] == prob['zlag']
    if kind == 'resident':
        assert plan['resident'] == 1 and plan['fast'] == 4, plan
    if kind == 'wavelet-8192':
        assert plan['fast'] == 0, plan              # (zlag: not the hand-structured chain wave)
    assert_same(orc, gpu, trace=trace)
    gpu.close()


# ---- lean_demc_lag_kernel: the hand-laid-out loop of the cadence ----------------------------
@pytest.mark.parametrize('zlag', [1, 2, 4, 7])
@pytest.mark.parametrize('case', ['plain', 'resume', 'abort', 'tight', 'tiny-history', 'fx',
                                  'thin', 'fx-thin'])
def test_lean_demc_loop_of_the_lagged_cadence(eng, monkeypatch, case, zlag):
    nchains = 5 if case == 'tiny-history' else 12
    thinning = {'thin': 3, 'fx-thin': 4}.get(case, 1)
    if case in ('fx', 'fx-thin'):
        # degree-4 polynomial, a shared and a fixed coefficient, two-sided Gaussian priors
        prob = H.shared_problem(nchains=nchains, nsamples=nchains*90*thinning, sampler='demc',
                                thinning=thinning, zlag=zlag)
        prob['params'] = prob['params'][:4]
        for k in ('pstep', 'pmin', 'pmax', 'prior', 'priorlow', 'priorup'):
            prob[k] = prob[k][:4]
        prob['pstep'] = np.array([0.3, 1.0, 0.0, -2.0])     # p2 fixed, p3 <- p1
    else:
        prob = H.quad_problem(ndata=300, nchains=nchains, nsamples=nchains*90*thinning,
                              sampler='demc', thinning=thinning, zlag=zlag)
    if case == 'tight':
        prob.update(params=np.array([4.0, -2.5, 0.5]), pmin=np.array([3.5, -2.8, 0.45]),
                    pmax=np.array([6.5, -2.2, 0.56]))
    if case == 'tiny-history':
        prob.update(hsize=1)
    if case == 'abort':
        monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '21')
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['recoveries'] == 1
        assert_same(orc, gpu, trace=False)
        gpu.close()
        return
    if case == 'resume':
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 20, 21, 22, 24):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert plan['persistent'] > 0 and plan['recoveries'] == 0 and plan['fast'] == 2, plan
    assert plan['zlag'] == zlag
    assert_same(orc, gpu, trace=False)
    if case == 'tight':
        assert gpu.state()['outbounds'].sum() > 50
    gpu.close()


def test_headline_shape_in_the_lagged_cadence_full_size(eng):
    """C2's shape (256 chains x 1000 points, demc) with zlag = 4: 2048 refreshes bit-exact, and
    the loop does not wait for anybody (an order of magnitude of the rate is asserted; the
    figure itself is bench.py's business)."""
    prob = H.quad_problem(ndata=1000, nchains=256, nsamples=256*2048, sampler='demc', zlag=4)
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['fast'] == 2
    assert_same(orc, gpu, trace=False)
    gpu.close()


# ---- mc3hip_recycle: streaming runs ---------------------------------------------------------
@pytest.mark.parametrize('sampler,zlag', [('demc', 0), ('demc', 3), ('mrw', 0)])
def test_recycled_buffer_carries_the_run_on(eng, sampler, zlag):
    """After mc3hip_recycle(keep, seed) the context equals one that was GIVEN the kept rows as
    its first refreshes and the chains' states, under the new key: the oracle is put into
    that state by hand and both run on -- identical rows, counters and best fits."""
    prob = H.quad_problem(ndata=200, nchains=10, nsamples=10*70, sampler=sampler, zlag=zlag)
    if sampler == 'mrw':
        prob['pstep'] = np.array([0.05, 0.02, 0.003])
    keep = 1 + zlag
    gpu = eng.DeviceChains(**prob)
    orc = H.make_oracle(prob, plan=gpu.plan)
    Z0, lp0 = H.initial_population(orc)
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(nrefresh=40, nthreads=4)
    gpu.run(40)
    with pytest.raises(Exception, match='look'):
        gpu.recycle(keep=zlag, seed=1)                      # (too few rows kept)
    gpu.recycle(keep=keep, seed=99)
    assert gpu.refresh_done == keep
    # the oracle, by hand
    n, M0 = orc.nchains, orc.M0
    for a in (orc.Z, orc.log_post, orc.zchain):
        a[M0:M0 + keep*n] = a[M0 + (40 - keep)*n:M0 + 40*n]
    orc.Z[M0 + keep*n:] = 0.0
    orc.log_post[M0 + keep*n:] = 0.0
    orc.zchain[M0 + keep*n:] = -1
    orc.S.refresh_done = keep
    orc.P.seed = 99
    orc.run(nrefresh=30, nthreads=4)
    gpu.run(30)
    Z, lp, zc = gpu.rows(0, M0 + (keep + 30)*n)
    np.testing.assert_array_equal(Z, orc.Z[:M0 + (keep + 30)*n])
    np.testing.assert_array_equal(lp, orc.log_post[:M0 + (keep + 30)*n])
    np.testing.assert_array_equal(zc, orc.zchain[:M0 + (keep + 30)*n])
    st = gpu.state()
    np.testing.assert_array_equal(st['numaccept'], orc.numaccept)
    np.testing.assert_array_equal(st['chisq'], orc.chisq)
    # rows behind the recycled ones are unwritten again
    Zr, _, zr = gpu.rows(M0 + (keep + 30)*n, n)
    assert np.all(np.isnan(Zr)) and np.all(zr == -1)
    gpu.close()


def test_snooker_history_is_not_recycled(eng):
    prob = H.quad_problem(nchains=8, nsamples=8*40, sampler='snooker')
    with eng.DeviceChains(**prob) as gpu:
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        gpu.set_population(Z0, lp0)
        gpu.run(20)
        with pytest.raises(Exception, match='whole history'):
            gpu.recycle(keep=2, seed=5)


# ---- lean_demc_kernel EPS: the headline loop with the support distribution -------------------
@pytest.mark.parametrize('case', ['plain', 'fx', 'resume', 'abort', 'tight', 'degree2', 'degree4'])
def test_headline_loop_with_the_support_distribution(eng, monkeypatch, case):
    """fepsilon != 0 (mc3/chain.py:185,232) on the hand-laid-out DEMC loop: lane k's normal
    deviate of the next refresh is drawn in the shadow of the partner rows' round trip."""
    nchains = 12
    if case == 'fx':
        prob = H.lean_shape_problem(nchains, 90, ndata=300, fx=1, fepsilon=0.02)
    elif case == 'degree2':
        prob = H.quad_problem(ndata=300, nchains=nchains, nsamples=nchains*90, sampler='demc',
                              fepsilon=0.01)
        for k in ('params', 'pstep', 'pmin', 'pmax'):
            prob[k] = prob[k][:2]
    elif case == 'degree4':
        prob = H.quad_problem(ndata=300, nchains=nchains, nsamples=nchains*90, sampler='demc',
                              fepsilon=0.01)
        prob.update(params=np.append(prob['params'], 0.0), pstep=np.append(prob['pstep'], 0.001),
                    pmin=np.append(prob['pmin'], -1.0), pmax=np.append(prob['pmax'], 1.0))
    else:
        prob = H.quad_problem(ndata=300, nchains=nchains, nsamples=nchains*90, sampler='demc',
                              fepsilon=0.01)
    if case == 'tight':
        prob.update(params=np.array([4.0, -2.5, 0.5]), pmin=np.array([3.5, -2.8, 0.45]),
                    pmax=np.array([6.5, -2.2, 0.56]))
    if case == 'abort':
        monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '21')
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['recoveries'] == 1
        assert_same(orc, gpu, trace=False)
        gpu.close()
        return
    if case == 'resume':
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 20, 21, 22, 24):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert plan['persistent'] > 0 and plan['recoveries'] == 0 and plan['fast'] == 2, plan
    assert_same(orc, gpu, trace=False)
    gpu.close()


def test_support_distribution_beyond_the_co_resident_waves(eng, monkeypatch):
    """... and an ensemble of such chains larger than the co-resident grid takes its turns on
    the general chain (no lean instance has the flavour): still exact."""
    monkeypatch.setenv('MC3HIP_PERSIST_CAPACITY', '5')
    prob = H.quad_problem(ndata=300, nchains=16, nsamples=16*40, sampler='demc', fepsilon=0.01)
    orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert plan['persistent'] > 0 and plan['chains_per_workgroup'] > 1 and plan['fast'] == 0, plan
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('case', ['plain', 'fx', 'resume', 'abort', 'tight', 'degree2', 'degree4',
                                  'zlag'])
def test_snooker_loop_with_the_support_distribution(eng, monkeypatch, case):
    """fepsilon != 0 on the hand-laid-out snooker loop (mc3/chain.py:185,217): two more Philox
    slots per refresh in the batch of draws -- five lanes per refresh, twelve refreshes per
    batch -- the deviates gathered through the crossbar."""
    nchains = 12
    kw = dict(zlag=2) if case == 'zlag' else {}
    if case == 'fx':
        prob = H.lean_shape_problem(nchains, 90, ndata=300, sampler='snooker', fx=1,
                                    fepsilon=0.02)
    else:
        prob = H.quad_problem(ndata=300, nchains=nchains, nsamples=nchains*90, sampler='snooker',
                              fepsilon=0.01, **kw)
    if case == 'degree2':
        for k in ('params', 'pstep', 'pmin', 'pmax'):
            prob[k] = prob[k][:2]
    if case == 'degree4':
        prob.update(params=np.append(prob['params'], 0.0), pstep=np.append(prob['pstep'], 0.001),
                    pmin=np.append(prob['pmin'], -1.0), pmax=np.append(prob['pmax'], 1.0))
    if case == 'tight':
        prob.update(params=np.array([4.0, -2.5, 0.5]), pmin=np.array([3.5, -2.8, 0.45]),
                    pmax=np.array([6.5, -2.2, 0.56]))
    if case == 'abort':
        monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '21')
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['recoveries'] == 1
        assert_same(orc, gpu, trace=False)
        gpu.close()
        return
    if case == 'resume':
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 11, 12, 13, 51):           # (a batch holds twelve refreshes)
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert plan['persistent'] > 0 and plan['recoveries'] == 0 and plan['fast'] == 2, plan
    assert_same(orc, gpu, trace=False)
    gpu.close()


# ---- kernels_rows.h in the cadence: several chains per wave side by side ---------------------
@pytest.mark.parametrize('zlag', [1, 2, 5])
@pytest.mark.parametrize('turns', [2, 3, 4])
@pytest.mark.parametrize('case', ['demc', 'demc-priors', 'demc-tiny-history', 'snooker',
                                  'snooker-tiny-history', 'mrw', 'demc-resume'])
def test_rows_loops_of_the_lagged_cadence(eng, monkeypatch, case, turns, zlag):
    """lean_demc_rows_lag_kernel (the partner rows of the next refresh asked for a refresh ahead),
    lean_snooker_rows_kernel (its history ends zlag refreshes earlier), lean_mrw_rows_kernel:
    ensembles beyond the co-resident waves (MC3HIP_PERSIST_CAPACITY pretends so) in the declared
    cadence, against the oracle in the same cadence."""
    sampler = case.split('-')[0]
    nchains = 5*turns - (1 if 'tiny' in case else 0)          # (ragged: the last turn not full)
    monkeypatch.setenv('MC3HIP_PERSIST_CAPACITY', '5')
    prob = H.quad_problem(ndata=400, nchains=nchains, nsamples=nchains*60, sampler=sampler,
                          zlag=zlag)
    if 'priors' in case:
        prob.update(prior=np.array([4.0, 0.0, 0.0]), priorlow=np.array([0.5, 0.0, 0.0]),
                    priorup=np.array([0.7, 0.0, 0.0]), pstep=np.array([1.0, 0.0, 0.1]))
    if 'tiny' in case:
        prob.update(hsize=1)
    if 'resume' in case:
        monkeypatch.setenv('MC3HIP_PERSIST_REFS', '7')
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 20, 37):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert plan['zlag'] == zlag and plan['persistent'] > 0 and plan['fast'] == 2, plan
    assert plan['chains_per_workgroup'] == turns and plan['recoveries'] == 0, plan
    assert_same(orc, gpu, trace=False)
    gpu.close()
