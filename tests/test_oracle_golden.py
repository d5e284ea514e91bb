"""Pin the oracle (CPU restatement) to the reference.

kernel level : the reference's own known-answer tests (reference
               tests/test_stats.py:115-171, 183-220, 280-302) and vectors made
               from the reference's compiled C (tests/golden/kernel_vectors.npz)
chain level  : deterministic reference runs (tests/golden/seq_*.npz, made by
               tests/golden/make_golden.py) replayed by
               oracle/chain_sequential.py -- Z rows, chain ids, accept and
               out-of-bounds counters must be IDENTICAL, log-posteriors agree
               to 1e-12 (the reference's -ffast-math sum is reassociated).
"""
import os

import numpy as np
import pytest

from oracle import kernels as ok
from oracle import chain_sequential as cseq


# ---- the reference's own known answers -------------------------------------
def test_chisq_known_answer():
    data = np.array([1.1, 1.2, 0.9, 1.0])
    model = np.ones(4)
    uncert = np.full(4, 0.1)
    np.testing.assert_allclose(ok.chisq(model, data, uncert), 6.0, rtol=1e-14)
    params = np.array([2.5, 5.5])
    priors = np.array([2.0, 5.0])
    plow = np.array([0.0, 1.0])
    pup = np.array([0.0, 1.0])
    np.testing.assert_allclose(
        ok.chisq(model, data, uncert, params, priors, plow, pup), 6.25,
        rtol=1e-14)


def test_residuals_known_answer():
    data = np.array([1.1, 1.2, 0.9, 1.0])
    model = np.ones(4)
    uncert = np.full(4, 0.1)
    np.testing.assert_allclose(
        ok.residuals(model, data, uncert), [-1.0, -2.0, 1.0, 0.0], atol=1e-14)
    res = ok.residuals(model, data, uncert, np.array([2.5, 5.5]),
                       np.array([2.0, 5.0]), np.array([0.0, 1.0]),
                       np.array([0.0, 1.0]))
    np.testing.assert_allclose(res, [-1.0, -2.0, 1.0, 0.0, 0.5], atol=1e-14)


def test_dwt_chisq_known_answer():
    data = np.array([2.0, 0.0, 3.0, -2.0, -1.0, 2.0, 2.0, 0.0])
    model = np.ones(8)
    params = np.array([1.0, 0.1, 0.1])
    np.testing.assert_allclose(ok.dwt_chisq(model, data, params), 1693.22308882)
    priors = np.array([1.0, 0.2, 0.3])
    plow = np.array([0.0, 0.0, 0.1])
    pup = np.array([0.0, 0.0, 0.1])
    np.testing.assert_allclose(
        ok.dwt_chisq(model, data, params, priors, plow, pup),
        1697.2230888243134, rtol=1e-13)
    with pytest.raises(ValueError, match='at least three parameters'):
        ok.dwt_chisq(model, data, np.array([1.0, 0.1]))


def test_daub4_known_answer():
    expected_inverse = np.array([
        -0.0301851821, -0.0522822690, -0.0662912607, -0.0824674511,
        -0.0905555462, -0.1008108399, -0.1132333322, -0.1250751254,
        0.1325825215, 0.3180280110, 0.4312613433, 0.5638438647,
        0.1412513157, -0.1325825215, -0.2576576469, -0.4225925490,
        -0.1671021007, -0.0242642855, 0.0059208966, 0.0662912607,
        0.0140089918, -0.0080880952, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    expected_forward = np.array([
        0.1625300592, 0.0874699408, -0.0463140877, 0.2795672632,
        -0.0905555462, 0.0, 0.0140089918, 0.1412513157, 0.3537658774,
        -0.0625, 0, 0, 0, 0, 0, -0.1082531755, 0, 0.8365163037,
        -0.1294095226, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    e4 = np.zeros(32)
    e4[4] = 1.0
    inv = ok.dwt_daub4(e4, True)
    np.testing.assert_allclose(inv, expected_inverse, atol=1e-9)
    np.testing.assert_allclose(ok.dwt_daub4(e4), expected_forward, atol=1e-9)
    np.testing.assert_allclose(ok.dwt_daub4(inv), e4, atol=1e-8)


def test_log_prior_known_answer():
    post = np.array([[3.0, 2.0], [3.1, 1.0], [3.6, 1.5]])
    z2 = np.zeros(2)
    np.testing.assert_equal(
        ok.log_prior(post, np.array([3.5, 0.0]), z2, z2, np.ones(2)),
        np.zeros(3))
    lo = np.array([0.1, 0.0])
    np.testing.assert_allclose(
        ok.log_prior(post, np.array([3.5, 0.0]), lo, lo, np.ones(2)),
        [-12.5, -8.0, -0.5])
    lo3 = np.array([0.1, 0.0, 0.0])
    np.testing.assert_allclose(
        ok.log_prior(post, np.array([3.5, 0.0, 0.0]), lo3, lo3,
                     np.array([1.0, 0.0, 1.0])), [-12.5, -8.0, -0.5])
    np.testing.assert_allclose(
        ok.log_prior(np.array([3.0, 2.0]), np.array([3.5, 0.0]), lo, lo,
                     np.ones(2)), -12.5)


# ---- vectors generated from the reference's compiled C ---------------------
def quad(p, x):
    return p[0] + p[1]*x + p[2]*x**2.0


def tutorial_data(ndata, seed):
    rs = np.random.RandomState(seed)
    x = np.linspace(0, 10, ndata)
    y = quad([3.0, -2.4, 0.5], x)
    uncert = np.sqrt(np.abs(y))
    return x, y + rs.normal(0, uncert), uncert


@pytest.fixture(scope='module')
def kv(golden_dir):
    return np.load(os.path.join(golden_dir, 'kernel_vectors.npz'))


@pytest.mark.parametrize('ndata', [8, 32, 1000, 16384])
def test_gauss_vectors(kv, ndata):
    x, data, uncert = tutorial_data(ndata, 1000 + ndata)
    P = kv[f'gauss_{ndata}_params']
    prior, plow, pup = kv['gauss_prior'], kv['gauss_priorlow'], kv['gauss_priorup']
    c0 = [ok.chisq(quad(p, x), data, uncert) for p in P]
    c1 = [ok.chisq(quad(p, x), data, uncert, p, prior, plow, pup) for p in P]
    np.testing.assert_allclose(c0, kv[f'gauss_{ndata}_chisq'], rtol=1e-12)
    np.testing.assert_allclose(c1, kv[f'gauss_{ndata}_chisq_priors'], rtol=1e-12)
    res = ok.residuals(quad(P[0], x), data, uncert, P[0], prior, plow, pup)
    np.testing.assert_allclose(res, kv[f'gauss_{ndata}_residuals0'], rtol=1e-13)


@pytest.mark.parametrize('ndata', [2, 4, 8, 32, 1024, 16384])
def test_dwt_vectors(kv, ndata):
    rs2 = np.random.RandomState(2000 + ndata)
    x = np.linspace(0, 10, ndata)
    data = quad([3.0, -2.4, 0.5], x) + rs2.normal(0, 1.0, ndata) \
        + 0.5*np.sin(2*np.pi*x)
    P = kv[f'dwt_{ndata}_params']
    prior, plow, pup = kv['dwt_prior'], kv['dwt_priorlow'], kv['dwt_priorup']
    c0 = [ok.dwt_chisq(quad(p[0:3], x), data, p) for p in P]
    c1 = [ok.dwt_chisq(quad(p[0:3], x), data, p, prior, plow, pup) for p in P]
    np.testing.assert_allclose(c0, kv[f'dwt_{ndata}_chisq'], rtol=1e-11)
    np.testing.assert_allclose(c1, kv[f'dwt_{ndata}_chisq_priors'], rtol=1e-11)


@pytest.mark.parametrize('n', [4, 8, 32, 100, 1024])
def test_daub4_vectors(kv, n):
    v = kv[f'daub4_{n}_in']
    np.testing.assert_allclose(ok.dwt_daub4(v), kv[f'daub4_{n}_fwd'],
                               rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(ok.dwt_daub4(v, True), kv[f'daub4_{n}_inv'],
                               rtol=1e-12, atol=1e-13)


def test_log_prior_vectors(kv):
    got = ok.log_prior(kv['logprior_post'], np.array([3.0, 0.0, 0.5]),
                       np.array([0.5, 0.0, 0.02]), np.array([0.25, 0.0, 0.04]),
                       np.array([1.0, 0.0, 0.1]))
    np.testing.assert_allclose(got, kv['logprior_value'], rtol=1e-14)


def test_ref_build_matches_restatement(kv):
    """When oracle/_ref (the reference's own C, compiled from its sources) is
    present, the restatement must agree with it directly."""
    mods = ok.load_ref()
    if mods is None:
        pytest.skip('oracle/_ref not built')
    cs, dwt = mods
    x, data, uncert = tutorial_data(1000, 2000)
    for p in kv['gauss_1000_params']:
        m = quad(p, x)
        np.testing.assert_allclose(ok.chisq(m, data, uncert),
                                   cs.chisq(m, data, uncert), rtol=1e-13)
    xw = np.linspace(0, 10, 1024)
    dw = np.random.RandomState(5).normal(0, 1, 1024)
    for p in kv['dwt_1024_params']:
        m = quad(p[0:3], xw)
        np.testing.assert_allclose(ok.dwt_chisq(m, dw, p),
                                   dwt.chisq(p, m, dw), rtol=1e-12)


# ---- chain level: replay deterministic reference runs ----------------------
def line_shared(p, x):
    return p[0] + p[1]*x + p[2]*x**2.0 + (p[3] - p[1])*x + (p[4] - 1.0)


CASES = {
    'quad_snooker': quad, 'quad_demc': quad, 'quad_mrw': quad,
    'quad_demc_eps_thin': quad, 'shared_priors_snooker': line_shared,
    'wavelet_snooker': quad, 'quad_demc_priors': quad, 'quad_mrw_bounds': quad,
}


def replay(g, func, chisq_fn=None):
    return cseq.run_sequential(
        int(g['hub_seed']), int(g['chain_seed']),
        data=g['data'], uncert=g['uncert'], func=func, params=g['params'],
        indparams=[g['x']], pmin=g['pmin'], pmax=g['pmax'], pstep=g['pstep'],
        prior=g['prior'], priorlow=g['priorlow'], priorup=g['priorup'],
        nchains=int(g['nchains']), nsamples=float(g['nsamples']),
        sampler=str(g['sampler']), wlike=bool(g['wlike']),
        thinning=int(g['thinning']), fgamma=float(g['fgamma']),
        fepsilon=float(g['fepsilon']), hsize=int(g['hsize']),
        kickoff=str(g['kickoff']), chisq_fn=chisq_fn)


@pytest.mark.parametrize('name', sorted(CASES))
def test_sequential_replay(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f'seq_{name}.npz'))
    run = replay(g, CASES[name])
    assert run.zsize == run.zlen
    # update_output keeps only the rows the chains wrote (zchain >= 0,
    # mc3/stats/stats.py:816-825); the initial population is dropped.
    v = run.zchain >= 0
    assert np.sum(~v) == run.M0
    np.testing.assert_array_equal(run.zchain[v], g['out_zchain'])
    np.testing.assert_array_equal(run.Z[v], g['out_posterior'])
    assert run.numaccept == int(g['numaccept'])
    np.testing.assert_array_equal(run.outbounds, g['outbounds'])
    np.testing.assert_array_equal(run.chainsize, g['chainsize'])
    np.testing.assert_allclose(run.log_post[v], g['out_log_post'], rtol=1e-12)
    np.testing.assert_array_equal(run.bestp, g['out_bestp'])
    np.testing.assert_allclose(run.best_log_post, g['out_best_log_post'],
                               rtol=1e-12)


@pytest.mark.parametrize('name', ['quad_demc', 'wavelet_snooker'])
def test_sequential_replay_with_ref_kernels(golden_dir, name):
    """Same replay with chisq taken from the reference's own compiled C
    (oracle/_ref): then even the log-posteriors are bit-identical."""
    fn = cseq.ref_chisq_fn()
    if fn is None:
        pytest.skip('oracle/_ref not built')
    g = np.load(os.path.join(golden_dir, f'seq_{name}.npz'))
    run = replay(g, CASES[name], chisq_fn=fn)
    v = run.zchain >= 0
    np.testing.assert_array_equal(run.Z[v], g['out_posterior'])
    np.testing.assert_array_equal(run.log_post[v], g['out_log_post'])
    assert run.numaccept == int(g['numaccept'])


def test_log_prior_with_a_log_uniform_parameter(golden_dir):
    """mc3/stats/stats.py:287-392 with all three kinds of prior (two-sided Gaussian,
    log-uniform where priorlow < 0, uniform) around a fixed parameter."""
    kv = np.load(os.path.join(golden_dir, 'residue_vectors.npz'))
    got = ok.log_prior(kv['logprior2_post'], kv['logprior2_prior'], kv['logprior2_priorlow'],
                       kv['logprior2_priorup'], kv['logprior2_pstep'])
    np.testing.assert_allclose(got, kv['logprior2_value'], rtol=1e-13)
    assert np.all(kv['logprior2_value'] < 0)


def test_design_matrix_chisq_at_full_c5_size(golden_dir):
    """BASELINE config 5's model (1e5 points x 10 columns): the oracle's chisq -- untiled and
    in the summation orders of the tiled plans -- against the reference's _chisq."""
    import helpers as H
    from oracle import lockstep as ls
    kv = np.load(os.path.join(golden_dir, 'residue_vectors.npz'))
    prob = H.linear_problem(ndata=100000, nmodel=10, nchains=16, nsamples=16)
    np.testing.assert_allclose([prob['data'].sum(), prob['indep'].sum()],
                               kv['c5_data_checksum'], rtol=1e-12)
    for extra in ({}, {'tile': 2048, 'vlanes': 1024, 'tile_tree': 1},
                  {'tile': 2048, 'vlanes': 1024, 'tile_tree': 2},
                  {'tile': 2048, 'vlanes': 1024, 'tile_tree': 0}):
        run = ls.LockstepRun(**prob, **extra)
        np.testing.assert_allclose(run.eval_chisq(kv['c5_params']), kv['c5_chisq_priors'],
                                   rtol=1e-10)
    noprior = dict(prob, prior=None, priorlow=None, priorup=None)
    np.testing.assert_allclose(ls.LockstepRun(**noprior).eval_chisq(kv['c5_params']),
                               kv['c5_chisq'], rtol=1e-10)
