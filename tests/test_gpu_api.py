"""mc3_amd.sample() on the GPU: identical output dictionary to the same call
with the CPU checker injected, and the reference's posterior (golden run of
mc3 v3.2.1) within Monte-Carlo error."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def quad(p, x):
    return p[0] + p[1]*x + p[2]*x**2.0


def both(checker_plan=None, **kw):
    import functools
    import mc3_amd
    from mc3_amd import utils as mu
    from oracle_engine import OracleChains
    outs = []
    checker = functools.partial(OracleChains, plan=checker_plan) if checker_plan else OracleChains
    for factory in (None, checker):
        np.random.seed(3)
        extra = {} if factory is None else {'engine_factory': factory}
        outs.append(mc3_amd.sample(log=mu.Log(verb=0), seed=21, **kw, **extra))
    return outs


# computed on the device in the GPU run and with NumPy in the checker's: reductions in
# another order (mean, std), another exp (KDE, log-uniform priors)
REDUCED = ('meanp', 'stdp', 'chisq', 'mode', 'hpd_low_bounds', 'hpd_high_bounds', 'CRlo',
           'CRhi')


def assert_same_dict(gpu, cpu):
    assert set(gpu) == set(cpu)
    for key in gpu:
        if key in ('pnames', 'texnames'):
            continue
        if key in REDUCED:
            np.testing.assert_allclose(np.asarray(gpu[key]), np.asarray(cpu[key]),
                                       rtol=1e-10, atol=1e-13, err_msg=key)
        else:
            np.testing.assert_array_equal(np.asarray(gpu[key]), np.asarray(cpu[key]),
                                          err_msg=key)


@pytest.mark.parametrize('sampler', ['snooker', 'demc', 'mrw'])
def test_sample_gpu_equals_cpu_checker(sampler, golden_dir):
    x, data, uncert = H.tutorial_data(100)
    gpu, cpu = both(data=data, uncert=uncert, func=quad,
                    params=np.array([10.0, -2.0, 0.1]), indparams=[x],
                    pstep=np.array([1.0, 0.5, 0.1]),
                    pmin=np.array([-10.0, -20.0, -10.0]),
                    pmax=np.array([40.0, 20.0, 10.0]),
                    sampler=sampler, nsamples=2100, burnin=20, nchains=7)
    assert_same_dict(gpu, cpu)
    if sampler == 'snooker':
        g = np.load(os.path.join(golden_dir, 'seq_quad_snooker.npz'))
        ref = g['out_posterior'][g['out_zmask']]
        post = gpu['posterior'][gpu['zmask']]
        assert np.all(np.abs(post.mean(0) - ref.mean(0)) < 0.5*ref.std(0))
        np.testing.assert_allclose(gpu['best_chisq'], g['out_best_chisq'], rtol=0.02)


def test_sample_gpu_shared_priors_wavelet_linear():
    import mc3_amd
    prob = H.shared_problem()
    poly5 = mc3_amd.models.Polynomial(5)
    gpu, cpu = both(data=prob['data'], uncert=prob['uncert'], func=poly5,
                    params=prob['params'], indparams=[prob['indep']],
                    pstep=prob['pstep'], pmin=prob['pmin'], pmax=prob['pmax'],
                    prior=prob['prior'], priorlow=prob['priorlow'],
                    priorup=prob['priorup'], sampler='snooker', nsamples=1600,
                    burnin=10, nchains=8, kickoff='uniform')
    assert_same_dict(gpu, cpu)
    prob = H.wavelet_problem()
    gpu, cpu = both(data=prob['data'], uncert=prob['uncert'], func=quad,
                    params=prob['params'], indparams=[prob['indep']],
                    pstep=prob['pstep'], pmin=prob['pmin'], pmax=prob['pmax'],
                    sampler='snooker', nsamples=1400, burnin=10, nchains=7, wlike=True)
    assert_same_dict(gpu, cpu)
    prob = H.linear_problem(nsamples=1600)
    gpu, cpu = both(data=prob['data'], uncert=prob['uncert'],
                    func=lambda p, A: p @ A, params=prob['params'],
                    indparams=[prob['indep']], pstep=prob['pstep'],
                    pmin=prob['pmin'], pmax=prob['pmax'], prior=prob['prior'],
                    priorlow=prob['priorlow'], priorup=prob['priorup'],
                    sampler='demc', nsamples=1600, burnin=10, nchains=16)
    assert_same_dict(gpu, cpu)


@pytest.mark.parametrize('sampler', ['demc', 'snooker'])
def test_sample_tiled_problem_equals_cpu_checker(sampler):
    """sample() end to end on a problem that takes the tiled plan (design matrix, 20 000
    points: the one-launch resident kernel): initial population, chains, Gelman-Rubin,
    posterior and best-fit statistics equal the checker's run of the same call, which sums
    the chain's proposals in the plan's order and arbitrary vectors (initial population,
    best fit) in the untiled order, as the library does."""
    prob = H.linear_problem(ndata=20000, nmodel=7)
    gpu, cpu = both(checker_plan={'tile': 2048, 'vlanes': 1024, 'tile_tree': 2},
                    data=prob['data'], uncert=prob['uncert'],
                    func=lambda p, A: p @ A, params=prob['params'],
                    indparams=[prob['indep']], pstep=prob['pstep'],
                    pmin=prob['pmin'], pmax=prob['pmax'], prior=prob['prior'],
                    priorlow=prob['priorlow'], priorup=prob['priorup'],
                    sampler=sampler, nsamples=32*60, burnin=10, nchains=32)
    assert_same_dict(gpu, cpu)


def test_sample_polynomial_of_six_coefficients_over_many_points():
    """A polynomial of five or more coefficients over many points reaches the device as the
    design matrix of the powers of x (mc3_amd.models.lower) and takes the one-launch tiled
    plan of the matrix cores; the output dictionary equals the checker's run of the same
    call."""
    import mc3_amd
    rs = np.random.RandomState(7)
    ndata = 20000
    x = np.linspace(-1.0, 1.0, ndata)
    truth = np.array([1.0, -0.5, 2.0, 0.3, -1.0, 0.7])
    uncert = np.full(ndata, 0.5)
    data = np.polyval(truth[::-1], x) + rs.normal(0.0, 0.5, ndata)
    seen = {}
    real = mc3_amd.engine.DeviceChains

    def spy(**kw):
        g = real(**kw)
        seen.update(model=kw['model'], plan=g.plan)
        return g
    kw = dict(data=data, uncert=uncert, func=mc3_amd.models.Polynomial(6),
              params=truth + 0.01, indparams=[x], pstep=np.full(6, 0.01),
              pmin=np.full(6, -10.0), pmax=np.full(6, 10.0), sampler='snooker',
              nsamples=32*50, burnin=10, nchains=32)
    gpu, cpu = both(checker_plan={'tile': 2048, 'vlanes': 1024, 'tile_tree': 2}, **kw)
    assert_same_dict(gpu, cpu)
    from mc3_amd import utils as mu
    np.random.seed(3)
    again = mc3_amd.sample(log=mu.Log(verb=0), seed=21, engine_factory=spy, **kw)
    assert seen['model'] == 'linear' and seen['plan']['resident'] == 1, seen
    np.testing.assert_array_equal(again['posterior'], gpu['posterior'])


@pytest.mark.parametrize('sampler,builder,zlag', [('snooker', H.quad_problem, 0),
                                                  ('demc', H.quad_problem, 0),
                                                  ('demc', H.linear_problem, 0),
                                                  ('demc', H.quad_problem, 2),
                                                  ('snooker', H.quad_problem, 1)])
def test_rccl_path_single_rank_in_a_graph(sampler, builder, zlag):
    """The RCCL exchange (one rank: the all-gather is a copy) captured in a hipGraph:
    step kernel -> ring slot, in-place ncclAllGather of the slot, scatter to Z, 16
    refreshes per replay; the first refresh and the tail run eagerly.  The multi-rank
    data flow is covered on CPU by test_distributed_gloo.py."""
    from mc3_amd import engine
    kw = dict(ndata=20000, nchains=16, nsamples=16*75) if builder is H.linear_problem \
        else dict(nchains=8, nsamples=8*75)
    prob = builder(sampler=sampler, **kw)
    os.environ['MC3HIP_GRAPH_NODES'] = '16'
    try:
        gpu = engine.DeviceChains(zlag=zlag, **prob)
    finally:
        del os.environ['MC3HIP_GRAPH_NODES']
    # (zlag > 0: the cadence on the one-launch-per-refresh path the all-gather rides on --
    # tests/test_gpu_multidevice.py runs it with a rank per GPU)
    orc = H.make_oracle(prob, plan=gpu.plan, zlag=zlag)
    Z0, lp0 = H.initial_population(orc)
    orc.set_population(Z0, lp0)
    orc.run(nthreads=4)
    with gpu:
        gpu.set_population(Z0, lp0)
        uid = engine.comm_unique_id()
        assert len(uid) == 128
        gpu.comm_init(uid, 0, 1)
        gpu.run(40)                 # 1 eager + 2 replays + 7 eager
        assert gpu.plan['comm_graph'] == 16, gpu.plan
        gpu.run()                   # 2 replays + 3 eager
        gpu.comm_gather_log_post()
        Z, lp, zc = gpu.rows()
        st = gpu.state()
    np.testing.assert_array_equal(Z, orc.Z)
    np.testing.assert_array_equal(lp, orc.log_post)
    np.testing.assert_array_equal(st['numaccept'], orc.numaccept)


def test_put_rows_host_staged_exchange():
    """Two contexts on one GPU, each owning half of the chains, exchanging rows
    through the host after every refresh == the unsharded run."""
    from mc3_amd import engine
    prob = H.quad_problem(nchains=8, nsamples=8*40, sampler='demc')
    orc = H.make_oracle(prob)
    Z0, lp0 = H.initial_population(orc)
    orc.set_population(Z0, lp0)
    orc.run()
    a = engine.DeviceChains(chain0=0, nlocal=4, **prob)
    b = engine.DeviceChains(chain0=4, nlocal=4, **prob)
    for g in (a, b):
        g.set_population(Z0, lp0)
    for R in range(orc.nzchain):
        a.run(1)
        b.run(1)
        r0 = orc.M0 + R*8
        Za, la, _ = a.rows(r0, 4)
        Zb, lb, _ = b.rows(r0 + 4, 4)
        a.put_rows(r0 + 4, Zb, lb)
        b.put_rows(r0, Za, la)
    for g in (a, b):
        Z, lp, zc = g.rows()
        np.testing.assert_array_equal(Z, orc.Z)
        np.testing.assert_array_equal(lp, orc.log_post)
    acc = np.concatenate([a.state()['numaccept'], b.state()['numaccept']])
    np.testing.assert_array_equal(acc, orc.numaccept)
    a.close()
    b.close()


def test_device_gelman_rubin_matches_host():
    """mc3hip_chain_moments + gelman_rubin_from_moments == the host statistic
    (mc3/stats/gelman.py restated in mc3_amd.stats.gelman_rubin) on the same rows."""
    from mc3_amd import engine, stats as ms
    prob = H.quad_problem(ndata=100, nchains=16, nsamples=16*3000, sampler='demc')
    orc = H.make_oracle(prob)
    Z0, lp0 = H.initial_population(orc)
    with engine.DeviceChains(**prob) as gpu:
        gpu.set_population(Z0, lp0)
        gpu.run(1234)
        Z, lp, zc = gpu.rows(0, gpu.M0 + 1234*16)
        for burn in (0, 100):
            mean, var = gpu.chain_moments(burn)
            got = ms.gelman_rubin_from_moments(mean, var, 1234 - burn)
            want = ms.gelman_rubin(Z, zc, burn)
            np.testing.assert_allclose(got, want, rtol=1e-11)
        with pytest.raises(Exception, match='leaves no rows'):
            gpu.chain_moments(1234)


def test_tutorial_config_against_reference_summary(golden_dir):
    """BASELINE.json configs[0]: examples/tutorial.py (quadratic fit, snooker, 7
    chains, 1e5 samples, burnin 1000).  The reference ran it with its own generator
    (tests/golden/tutorial_summary.npz); the GPU run must agree within Monte-Carlo
    error on every posterior summary and find the same best fit."""
    import mc3_amd
    from mc3_amd import utils as mu
    g = np.load(os.path.join(golden_dir, 'tutorial_summary.npz'))
    np.random.seed(314)
    x = np.linspace(0, 10, 1000)
    y = quad([3.0, -2.4, 0.5], x)
    uncert = np.sqrt(np.abs(y))
    data = y + np.random.normal(0, uncert)
    np.random.seed(2)
    out = mc3_amd.sample(
        data=data, uncert=uncert, func=quad, params=np.array([10.0, -2.0, 0.1]),
        indparams=[x], pmin=np.array([-10.0, -20.0, -10.0]),
        pmax=np.array([40.0, 20.0, 10.0]), pstep=np.array([1.0, 0.5, 0.1]),
        sampler='snooker', nsamples=1e5, burnin=1000, nchains=7, grtest=True,
        log=mu.Log(verb=0), seed=5)
    assert len(out['posterior']) == int(g['nrows'])
    sd = g['stdp']
    # ~1e5 samples with autocorrelation: means within 0.1 sigma, widths within 6 %
    assert np.all(np.abs(out['meanp'] - g['meanp']) < 0.1*sd)
    assert np.all(np.abs(out['medianp'] - g['medianp']) < 0.1*sd)
    assert np.all(np.abs(out['stdp']/sd - 1.0) < 0.06)
    assert np.all(np.abs(out['median_low_bounds'] - g['median_low_bounds']) < 0.15*sd)
    assert np.all(np.abs(out['median_high_bounds'] - g['median_high_bounds']) < 0.15*sd)
    # the best fit is a property of the data, not of the sampler
    np.testing.assert_allclose(out['best_chisq'], g['best_chisq'], rtol=2e-4)
    np.testing.assert_allclose(out['bestp'], g['bestp'], atol=0.3*sd.max(), rtol=0.05)
    np.testing.assert_allclose(out['stddev_residuals'], g['stddev_residuals'], rtol=1e-3)
    assert abs(out['acceptance_rate'] - g['acceptance_rate']) < 4.0


def test_resume_and_early_stop_gpu_equals_cpu_checker(tmp_path):
    """savefile -> resume (chains restart from their last Z rows, chain.py:166-169)
    and a Gelman-Rubin early stop (mcmc_driver.py:335-346), GPU vs CPU checker."""
    import mc3_amd
    from mc3_amd import utils as mu
    from oracle_engine import OracleChains
    x, data, uncert = H.tutorial_data(100)
    base = dict(data=data, uncert=uncert, func=quad, params=np.array([10.0, -2.0, 0.1]),
                indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
                pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
                sampler='demc', nchains=8, burnin=20, log=mu.Log(verb=0))
    outs = {}
    for name, extra in (('gpu', {}), ('cpu', {'engine_factory': OracleChains})):
        sf = str(tmp_path / f'{name}.npz')
        np.random.seed(4)
        first = mc3_amd.sample(nsamples=1600, savefile=sf, seed=1, **base, **extra)
        np.random.seed(5)
        more = mc3_amd.sample(nsamples=800, savefile=sf, resume=True, seed=2, **base, **extra)
        np.random.seed(6)
        stop = mc3_amd.sample(nsamples=80000, grbreak=1.05, grnmin=800, seed=3,
                              **dict(base, burnin=100), **extra)
        outs[name] = (first, more, stop)
    for g, c in zip(outs['gpu'], outs['cpu']):
        assert_same_dict(g, c)
    assert outs['gpu'][1]['posterior'].shape == (2400, 3)
    assert len(outs['gpu'][2]['posterior']) < 80000


def test_device_gelman_rubin_equals_the_reference(golden_dir):
    """The device reductions (mc3hip_chain_moments_rows -> gelman_rubin_from_moments) on
    the Z rows of the reference fixture against the reference's own gelman_rubin output
    (mc3/stats/gelman.py:12-92), 1e-10."""
    from mc3_amd import engine, stats as ms
    kv = np.load(os.path.join(golden_dir, 'kernel_vectors.npz'))
    Z, zc = kv['gr_Z'], kv['gr_zchain']
    nch = int(zc.max()) + 1
    M0 = int(np.sum(zc < 0))
    nit = (len(Z) - M0)//nch
    prob = H.quad_problem(nchains=nch, nsamples=nch*nit, sampler='demc', hsize=M0//nch)
    with engine.DeviceChains(**prob) as gpu:
        assert gpu.M0 == M0 and gpu.nfree == Z.shape[1]
        gpu.set_population(Z[:M0], np.zeros(M0))
        gpu.put_rows(M0, Z[M0:])
        for b, want in zip(kv['gr_burnins'], kv['gr_psrf']):
            mean, var = gpu.chain_moments(int(b), nrefresh=nit)
            got = ms.gelman_rubin_from_moments(mean, var, nit - int(b))
            np.testing.assert_allclose(got, want, rtol=1e-10)


@pytest.mark.parametrize('builder,kw,kickoff', [
    (H.quad_problem, dict(nchains=8, nsamples=80), 'normal'),
    (H.quad_problem, dict(nchains=8, nsamples=80), 'uniform'),
    (H.quad_problem, dict(ndata=1000, nchains=2048, nsamples=2048), 'normal'),   # C3: M0 = 20480
    (H.shared_problem, dict(), 'normal'),
    (H.shared_problem, dict(), 'uniform'),
    (H.quad_problem, dict(nchains=8, nsamples=80, pmin=np.array([9.0, -2.5, 0.0]),
                          pmax=np.array([10.5, -1.5, 0.2])), 'normal'),      # many rejections
    (H.wavelet_problem, dict(ndata=1024), 'normal'),
    (H.linear_problem, dict(), 'normal'),
])
def test_initial_population_on_the_device(builder, kw, kickoff):
    """mc3hip_init_population (trial draws, boundaries, shared copies, chisq and the
    in-order acceptance on the device; mc3/mcmc_driver.py:229-270) against the checker's
    restatement: identical rows, log-posteriors and counters."""
    from mc3_amd import engine
    prob = builder(**kw)
    with engine.DeviceChains(**prob) as gpu:
        orc = H.make_oracle(prob, plan=gpu.plan)
        for seed in (prob['seed'], 5):
            Z0, lp0, filled, failed = gpu.init_population(kickoff, seed)
            wZ, wlp, wfilled, wfailed = orc.init_population(kickoff, seed)
            assert (filled, failed) == (wfilled, wfailed)
            assert filled == gpu.M0
            np.testing.assert_array_equal(Z0, wZ)
            np.testing.assert_array_equal(lp0, wlp)


def test_initial_population_that_cannot_be_filled():
    """Boundaries that exclude (almost) every trial: 100 M0 failures, then the same
    error as the reference (mcmc_driver.py:264-270)."""
    import mc3_amd
    from mc3_amd import engine, utils as mu
    prob = H.quad_problem(nchains=4, nsamples=40, hsize=2)
    prob['pmin'] = np.array([9.9999, -2.00001, 0.09999])
    prob['pmax'] = np.array([10.0001, -1.99999, 0.10001])
    with engine.DeviceChains(**prob) as gpu:
        orc = H.make_oracle(prob, plan=gpu.plan)
        got = gpu.init_population('normal', 3)
        want = orc.init_population('normal', 3)
        assert got[2:] == want[2:] and got[3] == 100*gpu.M0 and got[2] < gpu.M0 - 1
        np.testing.assert_array_equal(got[0], want[0])
    x, data, uncert = H.tutorial_data(100)
    with pytest.raises(ValueError, match='Cannot populate an initial sample set'):
        mc3_amd.sample(data=data, uncert=uncert, func=quad, params=prob['params'],
                       indparams=[x], pstep=prob['pstep'], pmin=prob['pmin'],
                       pmax=prob['pmax'], sampler='demc', nsamples=400, nchains=4, hsize=2,
                       log=mu.Log(verb=0), seed=3)


def test_posterior_summary_on_the_device_equals_the_reference(golden_dir):
    """mc3hip_posterior_summary (reductions + exact radix selection where the rows are) and
    mc3hip_kde_pdf, through the statistics sample() builds from them, against the
    reference's calc_sample_statistics(calc_hpd=True) fixture (mc3/stats/stats.py:876-964)."""
    from mc3_amd import engine, stats as ms
    kv = np.load(os.path.join(golden_dir, 'kernel_vectors.npz'))
    post, bestp, pstep = kv['stats_posterior'], kv['stats_bestp'], kv['stats_pstep']
    want = kv['stats_result']
    nch, M0 = 10, 10
    nit = len(post)//nch
    prob = H.quad_problem(nchains=nch, nsamples=nch*nit, sampler='demc', hsize=1)
    with engine.DeviceChains(**prob) as gpu:
        gpu.set_population(post[:M0], np.zeros(M0))
        gpu.put_rows(M0, post)
        summary = ms.summarise(engine=gpu, burn=0, nrefresh=nit)
        host = ms.summarise(post)
        for key in ('median', 'low', 'high', 'min', 'max'):        # exact selection
            np.testing.assert_array_equal(summary[key], host[key], err_msg=key)
        np.testing.assert_allclose(summary['mean'], host['mean'], rtol=1e-13)
        np.testing.assert_allclose(summary['std'], host['std'], rtol=1e-12)
        got = ms.statistics_from_summary(summary, bestp, pstep)
        for g, w in zip(got, want[:5]):
            np.testing.assert_allclose(g, w, rtol=1e-12)
        mode, lo, hi = ms.hpd_statistics(post, np.arange(len(post)), summary, bestp, pstep,
                                         kde=gpu.kde_pdf)
        np.testing.assert_allclose(mode, want[5], rtol=1e-10)
        np.testing.assert_allclose(lo, want[6], rtol=1e-10)
        np.testing.assert_allclose(hi, want[7], rtol=1e-10)
        # a burn-in and other ranks: against NumPy on the same rows
        burned = post[7*nch:]
        s2 = gpu.posterior_summary(7, [0, 1, len(burned)//3, len(burned) - 1], nrefresh=nit)
        srt = np.sort(burned, axis=0)
        np.testing.assert_array_equal(s2['order'], srt[[0, 1, len(burned)//3, len(burned) - 1]])
        np.testing.assert_allclose(s2['var'], burned.var(axis=0), rtol=1e-12)


def test_kde_and_chisq_rows_on_the_device():
    from mc3_amd import engine, stats as ms
    import scipy.stats as ss
    rs = np.random.RandomState(12)
    sample = rs.gamma(2.0, 1.0, 50000)
    x = np.linspace(0.1, 8.0, 100)
    kernel = ss.gaussian_kde(sample)
    cov = np.var(sample, ddof=1)*len(sample)**(-0.4)
    np.testing.assert_allclose(engine.kde_pdf(sample, x, cov), kernel.evaluate(x), rtol=1e-10)
    # chisq of the rows = -2 (log_post - log_prior), two-sided Gaussian priors
    prob = H.shared_problem(nsamples=8*60)
    orc = H.make_oracle(prob)
    Z0, lp0 = H.initial_population(orc)
    with engine.DeviceChains(**prob) as gpu:
        gpu.set_population(Z0, lp0)
        gpu.run()
        Z, lp, zc = gpu.rows()
        want = -2.0*(lp - ms.log_prior(Z, gpu.prior, gpu.priorlow, gpu.priorup, gpu.pstep))
        np.testing.assert_array_equal(gpu.chisq_rows(), want)
        np.testing.assert_array_equal(gpu.chisq_rows(gpu.M0 + 3, 50), want[gpu.M0 + 3:gpu.M0 + 53])


def test_chisq_rows_against_the_reference_log_prior(golden_dir):
    """mc3hip_get_chisq (rows_chisq_kernel: -2 (log_post - log_prior), mc3/stats/stats.py:821
    with log_prior of :287-392) against the REFERENCE's log_prior -- a two-sided Gaussian, a
    log-uniform (priorlow < 0) and a uniform free parameter around a fixed one
    (tests/golden/residue_vectors.npz, made by tests/golden/make_golden.py residues)."""
    from mc3_amd import engine
    kv = np.load(os.path.join(golden_dir, 'residue_vectors.npz'))
    post, want_lp = kv['logprior2_post'], kv['logprior2_value']
    x, data, uncert = H.tutorial_data(50)
    nrows = len(post)
    lp0 = -0.5*np.random.RandomState(8).uniform(40.0, 90.0, nrows)
    with engine.DeviceChains(
            data=data, uncert=uncert, indep=x, params=np.array([3.0, -2.4, 1.0, 0.0]),
            pstep=kv['logprior2_pstep'], prior=kv['logprior2_prior'],
            priorlow=kv['logprior2_priorlow'], priorup=kv['logprior2_priorup'],
            nchains=nrows//10, nsamples=nrows, sampler='mrw', model='poly') as gpu:
        assert gpu.M0 == nrows and gpu.nfree == post.shape[1]
        gpu.set_population(post, lp0)
        got = gpu.chisq_rows(0, nrows)
    np.testing.assert_allclose(got, -2.0*(lp0 - want_lp), rtol=1e-12)
    # the prior terms themselves, each kind present and nonzero
    np.testing.assert_allclose(-0.5*got - lp0, -want_lp, rtol=1e-10)
    assert np.all(want_lp < -1e-3)


# ---- round 4: the lockstep semantics pinned to the reference at the headline shape ----------
def _agrees_with(out, g, tag, mean_tol=0.1, width_tol=0.05, acc_tol=2.0, tail_tol=0.2):
    sd = g[tag + '_stdp']
    post = np.asarray(out['posterior'])
    assert len(post) == int(g[tag + '_nrows'])
    assert np.all(np.abs(post.mean(0) - g[tag + '_meanp']) < mean_tol*sd), \
        (post.mean(0), g[tag + '_meanp'], sd)
    assert np.all(np.abs(post.std(0)/sd - 1.0) < width_tol), (post.std(0), sd)
    q = np.percentile(post, [2.5, 16, 50, 84, 97.5], axis=0)
    # (the tails of a finite sample move more than its centre: 0.1 sigma at the median and
    # the one-sigma quantiles, 0.2 sigma at the 2.5 / 97.5 % ones)
    tol = np.array([tail_tol, 0.1, 0.1, 0.1, tail_tol])[:, None]*sd
    assert np.all(np.abs(q - g[tag + '_quantiles']) < tol), (q, g[tag + '_quantiles'])
    assert abs(out['acceptance_rate'] - float(g[tag + '_acceptance_rate'])) < acc_tol, \
        (out['acceptance_rate'], g[tag + '_acceptance_rate'])
    # the best fit is a property of the data, not of the sampler
    np.testing.assert_allclose(out['best_chisq'], g[tag + '_best_chisq'], rtol=2e-4)


def test_headline_config_c2_against_the_reference_run(golden_dir):
    """BASELINE.json configs[1] at its own shape -- 256 chains, demc, ndata 1000 -- run by the
    reference itself (ncpu=1, 5e5 samples, burnin 500; tests/golden/make_golden.py headline:
    mc3/chain.py:183-299 as the reference runs it, chains one after another) against the
    GPU's lockstep run of the same call: posterior means within 0.1 sigma, widths within 5 %,
    quantiles, acceptance rate within 2 points, the same best chisq."""
    import mc3_amd
    from mc3_amd import utils as mu
    g = np.load(os.path.join(golden_dir, 'headline_summary.npz'))
    x, data, uncert = H.tutorial_data(1000)
    np.random.seed(5)
    out = mc3_amd.sample(
        data=data, uncert=uncert, func=quad, params=np.array([10.0, -2.0, 0.1]),
        indparams=[x], pmin=np.array([-10.0, -20.0, -10.0]),
        pmax=np.array([40.0, 20.0, 10.0]), pstep=np.array([1.0, 0.5, 0.1]),
        sampler='demc', nsamples=5e5, burnin=500, nchains=256, grtest=False,
        log=mu.Log(verb=0), seed=5)
    _agrees_with(out, g, 'c2')


def test_wavelet_snooker_run_against_the_reference_run(golden_dir):
    """The wavelet likelihood under snooker -- 64 chains, 2^10 points, 2e5 samples -- by the
    reference itself against the GPU's run of the same call (src_c/_dwt.c:56-119 inside
    mc3/chain.py:183-299)."""
    import mc3_amd
    from mc3_amd import utils as mu
    g = np.load(os.path.join(golden_dir, 'headline_summary.npz'))
    np.random.seed(6)
    out = mc3_amd.sample(
        data=g['wl_in_data'], uncert=g['wl_in_uncert'], func=quad, params=g['wl_in_params'],
        indparams=[g['wl_in_indparams']], pmin=g['wl_in_pmin'], pmax=g['wl_in_pmax'],
        pstep=g['wl_in_pstep'], sampler='snooker', nsamples=2e5, burnin=500, nchains=64,
        wlike=True, grtest=False, log=mu.Log(verb=0), seed=6)
    # (sigma_r has a long lower tail -- its 2.5 % quantile lies four standard deviations below
    # the mean -- which 2e5 correlated samples pin to no better than half a sigma)
    _agrees_with(out, g, 'wl', acc_tol=3.0, tail_tol=0.5)


def test_chisq_of_mid_run_rows_against_the_reference_kernels(golden_dir):
    """The direct 1e-10 leg at C2 and C4 sizes (as residue_vectors.npz is for C5):
      * 300 rows of the REFERENCE's own C2 run and 200 rows around C4's wavelet posterior,
        with the chisq its _chisq / _dwt gave them, through the device's evaluation;
      * rows the CHAIN KERNELS wrote mid-run at C2 and C4 size: -2 log_post against the
        reference-faithful restatement (oracle_kernels.c, itself pinned to the reference's
        compiled C at 1e-10 by tests/test_oracle_golden.py) -- no summation plan handed
        over: the checker adds the terms in the reference's order."""
    from mc3_amd import engine
    from oracle import kernels as ok
    g = np.load(os.path.join(golden_dir, 'headline_summary.npz'))
    # C2
    prob = H.quad_problem(ndata=1000, nchains=256, nsamples=256*300)
    with engine.DeviceChains(**prob) as gpu:
        got = gpu.eval_chisq(g['c2_rows'])
        np.testing.assert_allclose(got, g['c2_rows_chisq'], rtol=1e-10)
        Z0, lp0, filled, _ = gpu.init_population(seed=3)
        assert filled == gpu.M0
        gpu.set_population(Z0, lp0)
        gpu.run()
        assert gpu.plan['fast'] == 2                     # (lean_demc_kernel wrote these rows)
        Z, lp, _ = gpu.rows(gpu.M0 + 150*256, 256)
    want = [ok.chisq(quad(p, prob['indep']), prob['data'], prob['uncert']) for p in Z]
    np.testing.assert_allclose(-2.0*lp, want, rtol=1e-10)
    # C4
    prob = H.wavelet_problem(ndata=16384, nchains=256, nsamples=256*40)
    assert np.array_equal(g['c4_data_checksum'], [prob['data'].sum(), prob['indep'].sum()])
    with engine.DeviceChains(**prob) as gpu:
        got = gpu.eval_chisq(g['c4_rows'])
        np.testing.assert_allclose(got, g['c4_rows_chisq'], rtol=1e-10)
        Z0, lp0, filled, _ = gpu.init_population(seed=3)
        assert filled == gpu.M0
        gpu.set_population(Z0, lp0)
        gpu.run()
        assert gpu.plan['fast'] == 3                     # (the nine-wave kernel's chain wave)
        Z, lp, _ = gpu.rows(gpu.M0 + 20*256, 64)
        P = gpu.expand_free(Z)
    want = [ok.dwt_chisq(quad(p[0:3], prob['indep']), prob['data'], p) for p in P]
    np.testing.assert_allclose(-2.0*lp, want, rtol=1e-10)


@pytest.mark.parametrize('no_persist', [False, True])
def test_rows_of_confirmed_refreshes_are_read_while_the_next_segment_runs(no_persist):
    """mc3hip_get_rows_settled (round 6: what mcmc() copies to the host while the next segment
    computes): the rows a sync has confirmed, on a stream of the call's own -- equal to
    mc3hip_get_rows', refused beyond the confirmed refreshes, also for a context that launches
    once per refresh; and sample() end to end equals the run that pulls everything at the end."""
    from mc3_amd import engine, _lib
    prob = H.quad_problem(ndata=1000, nchains=64, nsamples=64*4000, sampler='demc')
    with engine.DeviceChains(no_persist=no_persist, **prob) as gpu:
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        n1 = 1500 if not no_persist else 40
        gpu.run(n1)                                   # (synchronised: confirmed)
        nrows = gpu.M0 + n1*64
        Z = np.empty((nrows, gpu.nfree)); lp = np.empty(nrows); zc = np.empty(nrows, np.int32)
        gpu.run(gpu.nzchain - n1 if not no_persist else 40, sync=False)     # in flight
        gpu.rows_settled_into(0, Z, lp, zc)
        with pytest.raises(_lib.Mc3HipError, match='confirmed'):
            gpu.rows_settled_into(nrows - 8, np.empty((16, gpu.nfree)), np.empty(16),
                                  np.empty(16, np.int32))
        gpu.sync()
        Z2, lp2, zc2 = gpu.rows(0, nrows)
        np.testing.assert_array_equal(Z, Z2)
        np.testing.assert_array_equal(lp, lp2)
        np.testing.assert_array_equal(zc, zc2)
        orc.run(nrefresh=n1, nthreads=8)
        np.testing.assert_array_equal(Z[gpu.M0:], orc.Z[gpu.M0:nrows])
