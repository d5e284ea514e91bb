"""Host side of the drop-in boundary on CPU: mc3_amd.sample()/mcmc() keep the
reference's signature, argument checks, error messages and output dictionary
(mc3/sampler_driver.py:25-601, mc3/mcmc_driver.py:18-378).  The chains are run
by the lockstep oracle injected through ``engine_factory`` (the GPU tests run
the same calls on the device and demand identical dictionaries)."""
import inspect
import os
import re

import numpy as np
import pytest

import mc3_amd
from mc3_amd import models, parallel
from mc3_amd import utils as mu
from oracle_engine import OracleChains
import helpers as H


def quad(p, x):
    return p[0] + p[1]*x + p[2]*x**2.0


x, data, uncert = H.tutorial_data(100)
params = np.array([10.0, -2.0, 0.1])
pstep = np.array([1.0, 0.5, 0.1])
pmin = np.array([-10.0, -20.0, -10.0])
pmax = np.array([40.0, 20.0, 10.0])
quiet = dict(engine_factory=OracleChains, seed=7)


def run(**kw):
    base = dict(data=data, uncert=uncert, func=quad, params=np.copy(params),
                indparams=[x], pstep=pstep, pmin=pmin, pmax=pmax,
                sampler='snooker', nsamples=2100, burnin=20, nchains=7,
                log=mu.Log(verb=0), **quiet)
    base.update(kw)
    np.random.seed(11)
    return mc3_amd.sample(**base)


def test_signature_matches_reference():
    """Names, order and defaults of mc3.sample (sampler_driver.py:25-38) and the
    32 positional arguments of mcmc (mcmc_driver.py:18-26)."""
    want = [
        ('data', None), ('uncert', None), ('func', None), ('params', None),
        ('indparams', []), ('indparams_dict', {}), ('pmin', None), ('pmax', None),
        ('pstep', None), ('prior', None), ('priorlow', None), ('priorup', None),
        ('sampler', None), ('ncpu', None), ('leastsq', None), ('chisqscale', False),
        ('nchains', 7), ('nsamples', None), ('burnin', 0), ('thinning', 1),
        ('grtest', True), ('grbreak', 0.0), ('grnmin', 0.5), ('wlike', False),
        ('fgamma', 1.0), ('fepsilon', 0.0), ('hsize', 10), ('kickoff', 'normal'),
        ('plots', False), ('theme', 'blue'), ('statistics', 'med_central'),
        ('ioff', False), ('showbp', True), ('savefile', None), ('resume', False),
        ('rms', False), ('log', None), ('pnames', None), ('texnames', None),
    ]
    sig = inspect.signature(mc3_amd.sample)
    got = [(n, p.default) for n, p in sig.parameters.items() if n != 'kwargs']
    assert got == want
    names = list(inspect.signature(mc3_amd.mcmc).parameters)
    assert names[:32] == [
        'data', 'uncert', 'func', 'params', 'indparams', 'indparams_dict',
        'pmin', 'pmax', 'pstep', 'prior', 'priorlow', 'priorup', 'nchains',
        'ncpu', 'nsamples', 'sampler', 'wlike', 'fit_output', 'grtest',
        'grbreak', 'grnmin', 'burnin', 'thinning', 'fgamma', 'fepsilon', 'hsize',
        'kickoff', 'savefile', 'resume', 'log', 'pnames', 'texnames']


def test_output_dictionary_matches_reference_layout(golden_dir):
    g = np.load(os.path.join(golden_dir, 'seq_quad_snooker.npz'))
    out = run()
    ref_keys = {k[4:] for k in g.files if k.startswith('out_')}
    assert ref_keys - set(out) == set()
    # ('zlag': which lockstep cadence produced the rows -- 0, the strict one, by default)
    assert set(out) - ref_keys == {'pnames', 'texnames', 'zlag'}
    assert out['zlag'] == 0
    for key in sorted(ref_keys):
        got, want = np.asarray(out[key]), g['out_' + key]
        assert got.shape == want.shape, key
        assert got.dtype.kind == want.dtype.kind, key
    # the initial population is drawn from the same NumPy stream as the
    # reference's (mcmc_driver.py:229-262): with equal seeds its best row is the
    # same row, so before any chain step the two runs share their starting point
    assert out['posterior'].shape == (2100, 3)
    assert np.all(out['zchain'] == np.tile(np.arange(7), 300))
    assert out['acceptance_rate'] > 5.0
    np.testing.assert_allclose(out['chisq'], -2*out['log_post'], rtol=1e-14)
    assert out['best_chisq'] == pytest.approx(-2*out['best_log_post'])
    assert out['best_chisq'] <= out['chisq'].min()*(1 + 1e-12)
    # same posterior as the reference within Monte-Carlo error
    ref_post = g['out_posterior'][g['out_zmask']]
    post = out['posterior'][out['zmask']]
    sd = ref_post.std(axis=0)
    assert np.all(np.abs(post.mean(axis=0) - ref_post.mean(axis=0)) < 0.5*sd)
    np.testing.assert_allclose(out['best_chisq'], g['out_best_chisq'], rtol=0.02)
    np.testing.assert_allclose(out['BIC'] - out['best_chisq'], 3*np.log(100))


def test_initial_population_equals_reference(golden_dir):
    """Z[0:M0] of a reference run and of mc3_amd with the same NumPy seed."""
    g = np.load(os.path.join(golden_dir, 'seq_quad_demc.npz'))
    from oracle import chain_sequential as cseq
    seq = cseq.SequentialRun(
        data=g['data'], uncert=g['uncert'], func=quad, params=g['params'],
        indparams=[g['x']], pmin=g['pmin'], pmax=g['pmax'], pstep=g['pstep'],
        nchains=8, nsamples=2400, sampler='demc')
    seq.init_population(np.random.RandomState(int(g['hub_seed'])))
    captured = {}

    class Spy(OracleChains):
        def set_population(self, Z0, lp0, X0=None, chisq0=None):
            captured['Z0'], captured['lp0'] = Z0.copy(), lp0.copy()
            super().set_population(Z0, lp0, X0, chisq0)

    np.random.seed(int(g['hub_seed']))
    mc3_amd.sample(data=g['data'], uncert=g['uncert'], func=quad,
                   params=g['params'], indparams=[g['x']], pstep=g['pstep'],
                   pmin=g['pmin'], pmax=g['pmax'], sampler='demc', nsamples=2400,
                   burnin=20, nchains=8, log=mu.Log(verb=0), engine_factory=Spy,
                   seed=1, kickoff_stream='numpy')
    np.testing.assert_array_equal(captured['Z0'], seq.Z[:seq.M0])
    np.testing.assert_allclose(captured['lp0'], seq.log_post[:seq.M0], rtol=1e-12)


def test_shared_and_fixed_parameters():
    """reference tests/test_mcmc.py:85-109: shared parameters equal their
    source, fixed ones keep their value, in every statistic."""
    def quad4(p, x):
        return p[0] + p[1]*x + p[2]*x**2.0 + 0.0*p[3]
    out = run(func=quad4, params=np.array([10.0, -2.0, 0.1, -2.0]),
              pstep=np.array([1.0, 0.5, 0.0, -2]),
              pmin=np.array([-10.0, -20, -10, -20]), pmax=np.array([40.0, 20, 10, 20]),
              sampler='demc', nchains=8, nsamples=2400)
    assert out['posterior'].shape[1] == 2
    for key in ('bestp', 'meanp', 'medianp', 'median_low_bounds', 'mode'):
        assert out[key][3] == out[key][1]
    assert out['bestp'][2] == 0.1 and out['stdp'][2] == 0.0
    assert out['CRlo'][2] == 0.0 and out['CRhi'][2] == 0.0


def test_model_recognition():
    low = models.lower(quad, params, [x], {}, 100, False, pstep)
    assert (low.family, low.ncoef, low.offset) == ('poly', 3, None)
    A = np.array([np.ones(100), np.sin(x), x**2])
    low = models.lower(lambda p, A: 1.5 + p @ A, params, [A], {}, 100, False, pstep)
    assert low.family == 'linear' and np.allclose(low.offset, 1.5)
    np.testing.assert_allclose(low.indep, A, rtol=1e-12, atol=1e-12)
    with pytest.raises(ValueError, match='not linear in its parameters'):
        models.lower(lambda p, x: p[0]*np.exp(p[1]*x), params[:2], [x], {}, 100,
                     False, pstep[:2])
    # five or more coefficients over many points: the powers of x as a design matrix (the
    # one-launch tiled plan of the matrix cores); few points, few coefficients or the wavelet
    # likelihood: a polynomial as ever
    xb = np.linspace(-1.0, 1.0, 9000)
    p6 = np.arange(1.0, 7.0)
    for func in (models.Polynomial(6), lambda p, x: np.polyval(p[::-1], x)):
        low = models.lower(func, p6, [xb], {}, 9000, False, np.ones(6))
        assert (low.family, low.ncoef) == ('linear', 6) and low.indep.shape == (6, 9000)
        np.testing.assert_array_equal(low.indep, np.array([xb**k for k in range(6)]))
    assert models.lower(models.Polynomial(6), p6, [xb[:100]], {}, 100, False,
                        np.ones(6)).family == 'poly'
    assert models.lower(models.Polynomial(4), p6[:4], [xb], {}, 9000, False,
                        np.ones(4)).family == 'poly'
    assert models.lower(models.Polynomial(6), np.arange(1.0, 10.0), [xb[:8192]], {}, 8192, True,
                        np.ones(9)).family == 'poly'
    out = run(func=models.Polynomial(3))
    ref = run()
    np.testing.assert_array_equal(out['posterior'], ref['posterior'])
    # an affine callable with an offset samples the same posterior as the
    # equivalent shifted data set
    out_off = run(func=lambda p, x: 2.0 + quad(p, x), data=data + 2.0)
    np.testing.assert_allclose(out_off['posterior'], ref['posterior'], rtol=1e-9)


def test_wavelet_likelihood_through_sample():
    prob = H.wavelet_problem()
    out = mc3_amd.sample(
        data=prob['data'], uncert=prob['uncert'], func=quad, params=prob['params'],
        indparams=[prob['indep']], pstep=prob['pstep'], pmin=prob['pmin'],
        pmax=prob['pmax'], sampler='snooker', nsamples=1400, burnin=10, nchains=7,
        wlike=True, log=mu.Log(verb=0), **quiet)
    assert out['posterior'].shape == (1400, 5)
    assert out['bestp'][3] == 1.0                   # gamma is fixed
    from oracle import kernels as ok
    best = ok.dwt_chisq(quad(out['bestp'][:3], prob['indep']), prob['data'],
                        out['bestp'])
    np.testing.assert_allclose(-2*out['best_log_post'], best, rtol=1e-10)


def test_savefile_resume_and_grbreak(tmp_path):
    sf = str(tmp_path / 'run.npz')
    first = run(savefile=sf, sampler='demc', nchains=8, nsamples=1600, grtest=True)
    saved = np.load(sf)
    np.testing.assert_array_equal(saved['posterior'], first['posterior'])
    assert float(saved['chisq_factor']) == 1.0
    more = run(savefile=sf, resume=True, sampler='demc', nchains=8, nsamples=800,
               seed=8)
    assert more['posterior'].shape == (1600 + 800, 3)
    np.testing.assert_array_equal(more['posterior'][:1600], first['posterior'])
    assert np.all(more['zchain'][1600:] == np.tile(np.arange(8), 100))
    # the resumed chains continue from their last states (chain.py:166-169)
    last = first['posterior'][-8:]
    moved = more['posterior'][1600:1608]
    assert np.sum(np.all(moved == last, axis=1)) >= 2      # rejected steps repeat
    stopped = run(sampler='demc', nchains=8, nsamples=80000, grbreak=1.05,
                  grnmin=800, burnin=100)
    assert len(stopped['posterior']) < 80000
    # the lockstep cadence is on record in the savefile, and a resumed run must continue the
    # same random process: another zlag is refused, the same one carries on
    assert int(np.load(sf)['zlag']) == 0 and first['zlag'] == 0
    with pytest.raises(ValueError, match='cadence zlag = 0, this run asks for zlag = 2'):
        run(savefile=sf, resume=True, sampler='demc', nchains=8, nsamples=800, seed=8, zlag=2)
    sf2 = str(tmp_path / 'lag.npz')
    lag = run(savefile=sf2, sampler='demc', nchains=8, nsamples=1600, zlag=2)
    assert lag['zlag'] == 2 and int(np.load(sf2)['zlag']) == 2
    assert not np.array_equal(lag['posterior'], first['posterior'])
    with pytest.raises(ValueError, match='cadence zlag = 2, this run asks for zlag = 0'):
        run(savefile=sf2, resume=True, sampler='demc', nchains=8, nsamples=800, seed=8)
    again = run(savefile=sf2, resume=True, sampler='demc', nchains=8, nsamples=800, seed=8,
                zlag=2)
    np.testing.assert_array_equal(again['posterior'][:1600], lag['posterior'])


# ---- the reference's error-path tests (tests/test_mcmc.py:328-409) ----------
def test_error_messages():
    base = dict(data=data, uncert=uncert, func=quad, params=np.copy(params),
                sampler='snooker', indparams=[x], pstep=pstep, nsamples=1e4,
                burnin=100, log=mu.Log(verb=0), **quiet)

    def without(key, **extra):
        kw = dict(base)
        kw.pop(key, None)
        kw.update(extra)
        return kw
    with pytest.raises(ValueError, match="'data' is a required argument"):
        mc3_amd.sample(**without('data'))
    with pytest.raises(ValueError, match="'uncert' is a required argument"):
        mc3_amd.sample(**without('uncert'))
    with pytest.raises(ValueError, match="'func' must be either a callable or an iterable"):
        mc3_amd.sample(**without('func'))
    with pytest.raises(ValueError, match="'params' is a required argument"):
        mc3_amd.sample(**without('params'))
    with pytest.raises(ValueError, match="'sampler' is a required argument"):
        mc3_amd.sample(**without('sampler'))
    with pytest.raises(ValueError, match="'nsamples' is a required argument for MCMC runs"):
        mc3_amd.sample(**without('nsamples'))
    msg = re.escape('The number of burned-in samples (2000) is greater than the '
                    'number of iterations per chain (1429)')
    with pytest.raises(ValueError, match=msg):
        mc3_amd.sample(**without('burnin', burnin=2000))
    msg = re.escape("Invalid 'leastsq' input (invalid). Must select from ['lm', 'trf']")
    with pytest.raises(ValueError, match=msg):
        mc3_amd.sample(**without('leastsq', leastsq='invalid'))
    with pytest.raises(ValueError, match='Some initial-guess values are out of bounds'):
        mc3_amd.sample(**without('pmax', pmax=np.array([5.0, 20.0, 10.0])))
    with pytest.raises(ValueError, match=r'size of the data array \(100\) does not '
                                         r'match the size of the func\(\) output \(50\)'):
        mc3_amd.sample(**without('indparams', indparams=[x[:50]]))
    # bounds that leave no room for the initial population
    with pytest.raises(ValueError, match='Cannot populate an initial sample set'):
        mc3_amd.sample(**without('pmax', pmin=np.array([9.9999999, -20, -10.0]),
                                 pmax=np.array([10.0000001, 20.0, 10.0])))
    # no CPU fallback: without an engine override the GPU library is required
    assert mc3_amd.mcmc.__defaults__[-1] is None


def test_burn_matches_reference_docstring_example():
    Z = np.expand_dims([0., 1, 10, 20, 30, 11, 31, 21, 12, 22, 32], axis=1)
    zchain = np.array([-1, -1, 0, 1, 2, 0, 2, 1, 0, 1, 2])
    post, zc, zmask = mu.burn({'posterior': Z, 'zchain': zchain, 'burnin': 1})
    assert list(post[:, 0]) == [11, 12, 21, 22, 31, 32]
    assert list(zmask) == [5, 8, 7, 9, 6, 10]
    post, zc, zmask = mu.burn({'posterior': Z, 'zchain': zchain, 'burnin': 1}, sort=False)
    assert list(post[:, 0]) == [11, 31, 21, 12, 22, 32]
    post, _, _ = mu.burn({'posterior': Z, 'zchain': zchain, 'burnin': 1}, burnin=0)
    assert list(post[:, 0]) == [10, 11, 12, 20, 21, 22, 30, 31, 32]


def test_initial_population_with_shared_parameters_equals_reference(golden_dir):
    """The hub tests bounds BEFORE refreshing the shared parameters, i.e. with the
    copies left by the previous in-bounds trial (mcmc_driver.py:249-255); uniform
    kickoff, tight bounds, one shared and one fixed parameter."""
    g = np.load(os.path.join(golden_dir, 'seq_shared_priors_snooker.npz'))

    def line_shared(p, x):
        return p[0] + p[1]*x + p[2]*x**2.0 + (p[3] - p[1])*x + (p[4] - 1.0)

    from oracle import chain_sequential as cseq
    seq = cseq.SequentialRun(
        data=g['data'], uncert=g['uncert'], func=line_shared, params=g['params'],
        indparams=[g['x']], pmin=g['pmin'], pmax=g['pmax'], pstep=g['pstep'],
        prior=g['prior'], priorlow=g['priorlow'], priorup=g['priorup'],
        nchains=8, nsamples=2400, sampler='snooker', kickoff='uniform')
    seq.init_population(np.random.RandomState(int(g['hub_seed'])))
    captured = {}

    class Spy(OracleChains):
        def set_population(self, Z0, lp0, X0=None, chisq0=None):
            captured['Z0'], captured['lp0'] = Z0.copy(), lp0.copy()
            super().set_population(Z0, lp0, X0, chisq0)

    np.random.seed(int(g['hub_seed']))
    out = mc3_amd.sample(
        data=g['data'], uncert=g['uncert'], func=line_shared, params=g['params'],
        indparams=[g['x']], pstep=g['pstep'], pmin=g['pmin'], pmax=g['pmax'],
        prior=g['prior'], priorlow=g['priorlow'], priorup=g['priorup'],
        sampler='snooker', nsamples=2400, burnin=10, nchains=8, kickoff='uniform',
        log=mu.Log(verb=0), engine_factory=Spy, seed=1, kickoff_stream='numpy')
    np.testing.assert_array_equal(captured['Z0'], seq.Z[:seq.M0])
    np.testing.assert_allclose(captured['lp0'], seq.log_post[:seq.M0], rtol=1e-10)
    # shared parameter follows its source, fixed one keeps its value
    assert out['bestp'][3] == out['bestp'][1] and out['bestp'][4] == 1.0


def test_gelman_rubin_equals_the_reference(golden_dir):
    """mc3_amd.stats.gelman_rubin (both its regular-layout fast path and the general
    one) and gelman_rubin_from_moments (what the device reductions feed) against
    outputs of the reference's own mc3.stats.gelman_rubin (mc3/stats/gelman.py:12-92;
    fixtures made by tests/golden/make_golden.py)."""
    from mc3_amd import stats as ms
    kv = np.load(os.path.join(golden_dir, 'kernel_vectors.npz'))
    Z, zc = kv['gr_Z'], kv['gr_zchain']
    nch = int(zc.max()) + 1
    M0 = int(np.sum(zc < 0))
    for b, want in zip(kv['gr_burnins'], kv['gr_psrf']):
        np.testing.assert_allclose(ms.gelman_rubin(Z, zc, int(b)), want, rtol=1e-12)
        chains = Z[M0:].reshape(-1, nch, Z.shape[1])[int(b):]
        got = ms.gelman_rubin_from_moments(chains.mean(axis=0), chains.var(axis=0),
                                           len(chains))
        np.testing.assert_allclose(got, want, rtol=1e-12)
    np.testing.assert_allclose(
        ms.gelman_rubin(kv['gr_irregular_Z'], kv['gr_irregular_zchain'], 5),
        kv['gr_irregular_psrf'], rtol=1e-12)


def test_numpy_stream_population_with_a_binding_shared_interval():
    """The reference tests the boundaries with the shared parameters' STALE copies
    (mcmc_driver.py:249-255).  When a shared parameter's interval is tighter than its
    source's that rule bites: the block-wise host path must then equal a literal
    trial-by-trial statement of the reference's loop."""
    from mc3_amd.mcmc_driver import numpy_stream_population
    prob = H.shared_problem()
    prob['pmax'] = prob['pmax'].copy()
    prob['pmax'][3] = -19.0                    # p3 <- p1, but p1 may go up to -15
    eng = OracleChains(**prob)
    M0 = eng.M0

    def literal(seed):
        rs = np.random.RandomState(seed)
        params, pstep, pmin, pmax = (prob[k] for k in ('params', 'pstep', 'pmin', 'pmax'))
        ifree, ishare = np.flatnonzero(pstep > 0), np.flatnonzero(pstep < 0)
        values = np.array(params, float)
        Z0, lp0 = np.zeros((M0, len(ifree))), np.zeros(M0)
        i = j = 0
        while i < M0 and j < 100*M0:
            trial = rs.normal(params[ifree], pstep[ifree])
            values[ifree] = trial
            if np.any(values > pmax) or np.any(values < pmin):
                j += 1
                continue
            for s in ishare:
                values[s] = values[-int(pstep[s]) - 1]
            lp = -0.5*eng.eval_chisq(values)
            if not np.isfinite(lp):
                j += 1
                continue
            Z0[i], lp0[i] = trial, lp
            i += 1
        return Z0, lp0, i

    want = literal(31)
    np.random.seed(31)
    got = numpy_stream_population(eng, prob['params'], prob['pstep'], prob['pmin'],
                                  prob['pmax'], M0, 'normal')
    assert got[2] == want[2]
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])

    # a share OF a share: p4 <- p3 <- p1.  The reference refreshes the shared parameters one
    # after another in index order (chain.py:246-247), so p4 takes p3's FRESH value
    prob['pstep'] = np.array([0.3, 1.0, 3.0, -2, -4])
    prob['pmin'] = np.array([2.0, -30.0, 30.0, -30.0, -30.0])
    prob['pmax'] = np.array([4.5, -15.0, 60.0, -15.0, -15.0])
    prob['params'] = np.array([3.2, -22.0, 45.0, -22.0, -22.0])
    eng = OracleChains(**prob)
    want = literal(32)
    np.random.seed(32)
    got = numpy_stream_population(eng, prob['params'], prob['pstep'], prob['pmin'],
                                  prob['pmax'], M0, 'normal')
    assert got[2] == want[2] == M0
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    # and the other order: p3 <- p4 <- p1 takes p4's value of the trial BEFORE
    prob['pstep'] = np.array([0.3, 1.0, 3.0, -5, -2])
    eng = OracleChains(**prob)
    want = literal(33)
    np.random.seed(33)
    got = numpy_stream_population(eng, prob['params'], prob['pstep'], prob['pmin'],
                                  prob['pmax'], M0, 'normal')
    assert got[2] == want[2]
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])

    # a FIXED parameter outside its boundaries fails every trial, as in the reference
    prob = H.shared_problem()
    prob['pmax'] = prob['pmax'].copy()
    prob['pmax'][4] = 0.5                      # p4 is fixed at 1.0
    eng = OracleChains(**prob)
    np.random.seed(34)
    got = numpy_stream_population(eng, prob['params'], prob['pstep'], prob['pmin'],
                                  prob['pmax'], 8, 'normal')
    assert got[2] == 0


def test_philox_population_is_the_default_and_deterministic():
    """Without kickoff_stream the initial population comes from the engine (device:
    mc3hip_init_population; here the checker's restatement), keyed by the seed alone."""
    captured = []

    class Spy(OracleChains):
        def set_population(self, Z0, lp0, X0=None, chisq0=None):
            captured.append((Z0.copy(), lp0.copy()))
            super().set_population(Z0, lp0, X0, chisq0)

    x, data, uncert = H.tutorial_data(100)
    for np_seed in (1, 2):
        np.random.seed(np_seed)              # must not matter
        mc3_amd.sample(data=data, uncert=uncert, func=quad, params=np.array([10.0, -2.0, 0.1]),
                       indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
                       pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
                       sampler='demc', nsamples=800, nchains=8, log=mu.Log(verb=0),
                       engine_factory=Spy, seed=77)
    np.testing.assert_array_equal(captured[0][0], captured[1][0])
    np.testing.assert_array_equal(captured[0][1], captured[1][1])
    Z0 = captured[0][0]
    assert Z0.shape == (80, 3) and np.all(np.isfinite(captured[0][1]))
    # around the initial guess with the proposal widths (chi-by-eye: 80 draws)
    assert np.all(np.abs(Z0.mean(axis=0) - [10.0, -2.0, 0.1]) < 4*np.array([1.0, 0.5, 0.1])/np.sqrt(80)*2)


def test_sample_statistics_equal_the_reference(golden_dir):
    """summarise -> statistics_from_summary + hpd_statistics (what sample() now calls, in
    their host form) against the reference's calc_sample_statistics(calc_hpd=True)
    (mc3/stats/stats.py:876-964) on a skewed posterior with a fixed and a shared
    parameter; percentiles_from_order against np.percentile bit for bit."""
    from mc3_amd import stats as ms
    kv = np.load(os.path.join(golden_dir, 'kernel_vectors.npz'))
    post, bestp, pstep = kv['stats_posterior'], kv['stats_bestp'], kv['stats_pstep']
    want = kv['stats_result']          # median, mean, std, low, high, mode, hpd low, hpd high
    summary = ms.summarise(post)
    got = ms.statistics_from_summary(summary, bestp, pstep)
    for g, w in zip(got, want[:5]):
        np.testing.assert_allclose(g, w, rtol=1e-12)
    zmask = np.arange(len(post))
    mode, lo, hi = ms.hpd_statistics(post, zmask, summary, bestp, pstep)
    np.testing.assert_allclose(mode, want[5], rtol=1e-10)
    np.testing.assert_allclose(lo, want[6], rtol=1e-10)
    np.testing.assert_allclose(hi, want[7], rtol=1e-10)
    srt = np.sort(post, axis=0)
    pct = [15.85, 50.0, 84.15, 0.0, 100.0, 33.333]
    np.testing.assert_array_equal(
        ms.percentiles_from_order(len(post), pct, lambda ranks: srt[ranks]),
        np.percentile(post, pct, axis=0))


def test_fit_matches_the_reference_known_answers():
    """mc3.fit's documented example (mc3/fit_driver.py:92-124; also the reference's
    tests/test_fit.py): same best fit from both optimisers, priors honoured, shared and
    fixed parameters kept."""
    def quad(p, x):
        return p[0] + p[1]*x + p[2]*x**2.0
    np.random.seed(10)
    x = np.linspace(0, 10, 100)
    y = quad([3.0, -2.4, 0.5], x)
    uncert = np.sqrt(np.abs(y))
    data = y + np.random.normal(0, uncert)
    params = np.array([10.0, -2.0, 0.1])
    lm = mc3_amd.fit(data, uncert, quad, np.copy(params), indparams=[x], leastsq='lm')
    trf = mc3_amd.fit(data, uncert, quad, np.copy(params), indparams=[x], leastsq='trf',
                      pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]))
    np.testing.assert_allclose(lm['bestp'], trf['bestp'], rtol=1e-6)
    # weighted linear least squares has a closed form
    A = np.vstack([np.ones_like(x), x, x**2]).T/uncert[:, None]
    want = np.linalg.lstsq(A, data/uncert, rcond=None)[0]
    np.testing.assert_allclose(lm['bestp'], want, rtol=1e-7)
    assert lm['best_chisq'] == pytest.approx(np.sum(((quad(want, x) - data)/uncert)**2), rel=1e-10)
    assert lm['best_log_post'] == pytest.approx(-0.5*lm['best_chisq'], rel=1e-12)
    np.testing.assert_allclose(lm['best_model'], quad(lm['bestp'], x))
    # a Gaussian prior pulls, a fixed parameter stays, a shared one follows its source
    pri = mc3_amd.fit(data, uncert, quad, np.copy(params), indparams=[x],
                      prior=np.array([3.5, 0.0, 0.0]), priorlow=np.array([0.01, 0.0, 0.0]),
                      priorup=np.array([0.01, 0.0, 0.0]))
    assert abs(pri['bestp'][0] - 3.5) < 0.02 and pri['best_chisq'] > lm['best_chisq']
    assert pri['best_log_post'] < -0.5*pri['best_chisq'] + 1e-9       # the prior's share
    fx = mc3_amd.fit(data, uncert, lambda p, x: p[0] + p[1]*x + p[2]*x**2 + 0*p[3],
                     np.array([10.0, -2.0, 0.1, -2.0]), indparams=[x],
                     pstep=np.array([1.0, 0.5, 0.0, -2]))
    assert fx['bestp'][2] == 0.1 and fx['bestp'][3] == fx['bestp'][1]
    with pytest.raises(ValueError, match="Invalid 'leastsq' input"):
        mc3_amd.fit(data, uncert, quad, params, indparams=[x], leastsq='invalid')


def test_sample_with_the_least_squares_prefit():
    """leastsq / chisqscale as in mc3/sampler_driver.py:411-473: the chains start from the
    optimiser's best fit, chisqscale rescales the uncertainties to a reduced chisq of 1
    (and does nothing without leastsq, as in the reference), chisq_factor is reported."""
    from oracle_engine import OracleChains
    x, data, uncert = H.tutorial_data(100)
    kw = dict(data=data, uncert=uncert*0.5, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
              indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
              pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
              sampler='snooker', nsamples=7*150, burnin=20, nchains=7, grtest=False,
              log=mu.Log(verb=0), engine_factory=OracleChains, seed=5)
    plain = mc3_amd.sample(**kw, chisqscale=True)           # no leastsq: a no-op
    assert plain['chisq_factor'] == 1.0
    fitted = mc3_amd.sample(**kw, leastsq='lm')
    ref = mc3_amd.fit(data, uncert*0.5, H.quad, np.array([10.0, -2.0, 0.1]), indparams=[x])
    assert fitted['chisq_factor'] == 1.0
    assert fitted['best_log_post'] >= ref['best_log_post'] - 1e-9
    assert fitted['best_chisq'] <= plain['best_chisq'] + 1e-9
    scaled = mc3_amd.sample(**kw, leastsq='trf', chisqscale=True)
    want = np.sqrt(ref['best_chisq']/(100 - 3))
    assert scaled['chisq_factor'] == pytest.approx(want, rel=1e-6)
    assert scaled['red_chisq'] == pytest.approx(1.0, abs=0.05)


def test_wavelet_series_that_is_no_power_of_two_is_padded_on_request():
    """sample(..., wlike=True) with 1000 points: refused without ``wpad`` (the reference's zero
    padding transforms the unpadded length, src_c/_dwt.c:82-94), padded with zero residuals to
    1024 with it -- the oracle engine stands in for the device; outputs keep the data's length."""
    import mc3_amd
    from mc3_amd import utils as mu
    from oracle_engine import OracleChains
    x, data, uncert = H.tutorial_data(1000)
    kw = dict(data=data, uncert=uncert, func=H.quad, indparams=[x],
              params=np.array([3.0, -2.4, 0.5, 1.0, 0.5, 1.0]),
              pstep=np.array([0.3, 0.1, 0.02, 0.0, 0.1, 0.1]),
              pmin=np.array([-10.0, -20.0, -10.0, 0.0, 0.01, 0.01]),
              pmax=np.array([40.0, 20.0, 10.0, 2.0, 10.0, 10.0]),
              sampler='snooker', nsamples=7*60, burnin=5, nchains=7, wlike=True, grtest=False,
              log=mu.Log(verb=0), seed=3, engine_factory=OracleChains, kickoff_stream='numpy')
    with pytest.raises(ValueError, match='power-of-two'):
        mc3_amd.sample(**kw)
    np.random.seed(11)
    out = mc3_amd.sample(wpad=True, **kw)
    assert out['best_model'].shape == (1000,) and np.isfinite(out['best_chisq'])
    assert out['posterior'].shape[1] == 5
    # the chi-squared the run reports for its best fit is the padded series' (zero residuals on
    # the padding), i.e. the reference-faithful kernel's on such residuals
    from oracle import kernels as K
    model = np.concatenate([out['best_model'], np.zeros(24)])
    padded = np.concatenate([data, np.zeros(24)])
    np.testing.assert_allclose(out['best_chisq'], K.dwt_chisq(model, padded, out['bestp']),
                               rtol=1e-9)
