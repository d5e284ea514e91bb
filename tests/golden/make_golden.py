#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference itself.

Runs ONLY in the build container, where /root/reference exists:

    make -C oracle ref            # the reference's C modules -> oracle/_ref
    python tests/golden/make_golden.py

It imports mc3 v3.2.1 straight from /root/reference (nothing is copied) with
oracle/_ref on sys.path for its compiled modules, makes the runs deterministic
the way SURVEY.md section 8c describes -- ``mc3.chain.random.randint`` replaced
by a constant (the reference reseeds NumPy from it inside the forked worker,
mc3/chain.py:180), ``np.random.seed`` set in the parent, ``ncpu=1`` -- and
stores inputs + outputs as small fixtures.  The fixtures are data only; no
reference source travels.
"""
import io
import os
import sys
import contextlib

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
sys.path.insert(0, '/root/reference')

import numpy as np  # noqa: E402
import mc3  # noqa: E402
import mc3.stats as ms  # noqa: E402
import mc3.chain  # noqa: E402

CHAIN_SEED = 4242


class _ConstRandom:
    """Stand-in for the ``random`` module inside mc3.chain (HZ1)."""
    @staticmethod
    def randint(a, b):
        return CHAIN_SEED


sys.modules['mc3.chain'].random = _ConstRandom


# ---- the draw tape -----------------------------------------------------------
# Every np.random call the worker makes (mc3/chain.py:185-259) is forwarded to the real
# generator and its result recorded: kind 0 = one element of the normal vector, 1 =
# randint, 2 = uniform.  The worker is a forked process, so the tape is written to a
# scratch file when Chain.run returns and picked up by the parent.
TAPE_PATH = '/tmp/mc3_golden_tape.npz'


class _RecordingRandom:
    def __init__(self):
        self.kinds, self.values = [], []

    def seed(self, s):
        np.random.seed(s)

    def normal(self, *a, **k):
        v = np.random.normal(*a, **k)
        for x in np.atleast_1d(v):
            self.kinds.append(0)
            self.values.append(float(x))
        return v

    def randint(self, *a, **k):
        v = np.random.randint(*a, **k)
        self.kinds.append(1)
        self.values.append(float(v))
        return v

    def uniform(self, *a, **k):
        v = np.random.uniform(*a, **k)
        self.kinds.append(2)
        self.values.append(float(v))
        return v


class _NumpyProxy:
    """``np`` as mc3.chain sees it: everything is NumPy's, except ``random``."""
    def __init__(self):
        self.random = _RecordingRandom()

    def __getattr__(self, name):
        return getattr(np, name)


_proxy = _NumpyProxy()
sys.modules['mc3.chain'].np = _proxy
_orig_run = sys.modules['mc3.chain'].Chain.run


def _recorded_run(self):
    _proxy.random.kinds.clear()
    _proxy.random.values.clear()
    try:
        _orig_run(self)
    finally:
        np.savez(TAPE_PATH, kind=np.array(_proxy.random.kinds, np.int8),
                 value=np.array(_proxy.random.values, np.float64))


sys.modules['mc3.chain'].Chain.run = _recorded_run


def quad(p, x):
    return p[0] + p[1]*x + p[2]*x**2.0


def line_shared(p, x):
    # 5 parameters: p3 is shared with p1 (pstep=-2), p4 is fixed
    return p[0] + p[1]*x + p[2]*x**2.0 + (p[3] - p[1])*x + (p[4] - 1.0)


def tutorial_data(ndata, seed=314):
    """examples/tutorial.py:22-29 at a chosen size."""
    rs = np.random.RandomState(seed)
    x = np.linspace(0, 10, ndata)
    p_true = [3.0, -2.4, 0.5]
    y = quad(p_true, x)
    uncert = np.sqrt(np.abs(y))
    data = y + rs.normal(0, uncert)
    return x, data, uncert


_captured = {}
_orig_update_output = ms.update_output


def _capture_update_output(output, chain, hsize):
    _captured['outbounds'] = np.array(chain.outbounds[:], dtype=np.int64)
    _captured['numaccept'] = int(chain.numaccept.value)
    _captured['chainsize'] = np.array(chain.chainsize, dtype=np.int64)
    # the initial population (mcmc_driver.py:229-270): rows the chains did not write
    Z = np.array(chain.Z).reshape(-1, chain.nfree)
    init = np.array(chain.zchain) < 0
    _captured['Z0'] = Z[init].copy()
    _captured['log_post0'] = np.array(chain.log_post)[init].copy()
    return _orig_update_output(output, chain, hsize)


sys.modules['mc3.mcmc_driver'].ms.update_output = _capture_update_output


def run_case(name, hub_seed, func, **kw):
    np.random.seed(hub_seed)
    cwd = os.getcwd()
    os.chdir('/tmp')
    if os.path.exists(TAPE_PATH):
        os.remove(TAPE_PATH)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out = mc3.sample(func=func, ncpu=1, plots=False, grtest=False,
                             log=None, **kw)
    finally:
        os.chdir(cwd)
    tape = np.load(TAPE_PATH)
    save = dict(
        tape_kind=tape['kind'], tape_value=tape['value'],
        hub_seed=hub_seed, chain_seed=CHAIN_SEED,
        data=kw['data'], uncert=kw['uncert'], x=kw['indparams'][0],
        params=kw['params'], pstep=kw['pstep'],
        pmin=kw.get('pmin', np.tile(-np.inf, len(kw['params']))),
        pmax=kw.get('pmax', np.tile(np.inf, len(kw['params']))),
        prior=kw.get('prior', np.zeros(len(kw['params']))),
        priorlow=kw.get('priorlow', np.zeros(len(kw['params']))),
        priorup=kw.get('priorup', np.zeros(len(kw['params']))),
        nchains=kw['nchains'], nsamples=kw['nsamples'],
        sampler=kw['sampler'], wlike=kw.get('wlike', False),
        thinning=kw.get('thinning', 1), burnin=kw.get('burnin', 0),
        fgamma=kw.get('fgamma', 1.0), fepsilon=kw.get('fepsilon', 0.0),
        hsize=kw.get('hsize', 10), kickoff=kw.get('kickoff', 'normal'),
        outbounds=_captured['outbounds'], numaccept=_captured['numaccept'],
        chainsize=_captured['chainsize'],
        Z0=_captured['Z0'], log_post0=_captured['log_post0'],
    )
    for key, val in out.items():
        if key in ('pnames', 'texnames'):
            continue
        save['out_' + key] = np.asarray(val)
    np.savez_compressed(os.path.join(HERE, f'seq_{name}.npz'), **save)
    print(f'{name}: numaccept={save["numaccept"]} outbounds={save["outbounds"]} '
          f'rows={len(out["posterior"])}')


def chain_cases():
    x, data, uncert = tutorial_data(100)
    base = dict(data=data, uncert=uncert, indparams=[x],
                params=np.array([10.0, -2.0, 0.1]),
                pstep=np.array([1.0, 0.5, 0.1]),
                pmin=np.array([-10.0, -20.0, -10.0]),
                pmax=np.array([40.0, 20.0, 10.0]))
    run_case('quad_snooker', 11, quad, sampler='snooker', nchains=7,
             nsamples=2100, burnin=20, **base)
    run_case('quad_demc', 12, quad, sampler='demc', nchains=8,
             nsamples=2400, burnin=20, **base)
    run_case('quad_mrw', 13, quad, sampler='mrw', nchains=7,
             nsamples=1400, burnin=10, **base)
    run_case('quad_demc_eps_thin', 14, quad, sampler='demc', nchains=9,
             nsamples=1800, thinning=2, fepsilon=0.01, fgamma=0.9, burnin=10,
             **base)
    # tight bounds (out-of-bounds counters), Gaussian priors (two-sided),
    # one shared and one fixed parameter, uniform kickoff:
    x, data, uncert = tutorial_data(64)
    run_case('shared_priors_snooker', 15, line_shared,
             data=data, uncert=uncert, indparams=[x],
             params=np.array([3.2, -2.2, 0.45, -2.2, 1.0]),
             pstep=np.array([0.3, 0.1, 0.03, -2, 0.0]),
             pmin=np.array([2.0, -3.0, 0.42, -3.0, 0.0]),
             pmax=np.array([4.5, -1.9, 0.56, -1.9, 2.0]),
             prior=np.array([3.0, 0.0, 0.5, 0.0, 0.0]),
             priorlow=np.array([0.5, 0.0, 0.02, 0.0, 0.0]),
             priorup=np.array([0.25, 0.0, 0.04, 0.0, 0.0]),
             sampler='snooker', nchains=8, nsamples=2400, burnin=10,
             kickoff='uniform')
    # DEMC with two-sided Gaussian priors on two parameters (+ support distribution):
    x, data, uncert = tutorial_data(100)
    run_case('quad_demc_priors', 17, quad, sampler='demc', nchains=8,
             nsamples=2400, burnin=20, fepsilon=0.005,
             prior=np.array([3.0, 0.0, 0.5]),
             priorlow=np.array([0.5, 0.0, 0.02]),
             priorup=np.array([0.25, 0.0, 0.04]), **base)
    # MRW inside tight boundaries: many out-of-bounds proposals (which consume no
    # acceptance draw, chain.py:241-243,259)
    run_case('quad_mrw_bounds', 18, quad, sampler='mrw', nchains=7,
             nsamples=2100, burnin=10,
             data=data, uncert=uncert, indparams=[x],
             params=np.array([4.0, -2.5, 0.5]),
             pstep=np.array([1.0, 0.5, 0.1]),
             pmin=np.array([3.5, -2.8, 0.45]),
             pmax=np.array([6.5, -2.2, 0.56]))
    # wavelet likelihood: quad + (gamma fixed, sigma_r, sigma_w), ndata = 2^6
    rs = np.random.RandomState(99)
    x = np.linspace(0, 10, 64)
    y = quad([3.0, -2.4, 0.5], x)
    data = y + rs.normal(0, 1.0, 64) + 0.5*np.sin(2*np.pi*x)
    uncert = np.ones(64)
    run_case('wavelet_snooker', 16, quad,
             data=data, uncert=uncert, indparams=[x],
             params=np.array([3.0, -2.4, 0.5, 1.0, 0.5, 1.0]),
             pstep=np.array([0.3, 0.1, 0.02, 0.0, 0.1, 0.1]),
             pmin=np.array([-10.0, -20.0, -10.0, 0.0, 0.01, 0.01]),
             pmax=np.array([40.0, 20.0, 10.0, 2.0, 10.0, 10.0]),
             sampler='snooker', nchains=7, nsamples=1400, burnin=10,
             wlike=True)


def kernel_vectors():
    """Standalone likelihood-kernel vectors: (params, data) -> chisq."""
    out = {}
    rs = np.random.RandomState(2718)
    for ndata in (8, 32, 1000, 16384):
        x, data, uncert = tutorial_data(ndata, seed=1000 + ndata)
        P = np.array([3.0, -2.4, 0.5]) + rs.normal(0, [0.3, 0.1, 0.02], (6, 3))
        prior = np.array([3.0, 0.0, 0.5])
        plow = np.array([0.5, 0.0, 0.02])
        pup = np.array([0.25, 0.0, 0.04])
        c0 = [ms.chisq(quad(p, x), data, uncert) for p in P]
        c1 = [ms.chisq(quad(p, x), data, uncert, p, prior, plow, pup) for p in P]
        out[f'gauss_{ndata}_params'] = P
        out[f'gauss_{ndata}_chisq'] = np.array(c0)
        out[f'gauss_{ndata}_chisq_priors'] = np.array(c1)
        out[f'gauss_{ndata}_residuals0'] = ms.residuals(
            quad(P[0], x), data, uncert, P[0], prior, plow, pup)
    out['gauss_prior'] = prior
    out['gauss_priorlow'] = plow
    out['gauss_priorup'] = pup
    for ndata in (2, 4, 8, 32, 1024, 16384):
        rs2 = np.random.RandomState(2000 + ndata)
        x = np.linspace(0, 10, ndata)
        data = quad([3.0, -2.4, 0.5], x) + rs2.normal(0, 1.0, ndata) \
            + 0.5*np.sin(2*np.pi*x)
        P = np.zeros((5, 6))
        P[:, 0:3] = np.array([3.0, -2.4, 0.5]) + rs.normal(0, [0.3, 0.1, 0.02], (5, 3))
        P[:, 3] = 1.0
        P[:, 4] = rs.uniform(0.1, 2.0, 5)
        P[:, 5] = rs.uniform(0.5, 2.0, 5)
        prior = np.array([3.0, 0.0, 0.5, 0.0, 0.5, 0.0])
        plow = np.array([0.5, 0.0, 0.0, 0.0, 0.2, 0.0])
        pup = np.array([0.25, 0.0, 0.0, 0.0, 0.1, 0.0])
        c0 = [ms.dwt_chisq(quad(p[0:3], x), data, p) for p in P]
        c1 = [ms.dwt_chisq(quad(p[0:3], x), data, p, prior, plow, pup) for p in P]
        out[f'dwt_{ndata}_params'] = P
        out[f'dwt_{ndata}_chisq'] = np.array(c0)
        out[f'dwt_{ndata}_chisq_priors'] = np.array(c1)
    out['dwt_prior'] = prior
    out['dwt_priorlow'] = plow
    out['dwt_priorup'] = pup
    # D4 transform vectors (forward / inverse), power-of-two and padded sizes
    for n in (4, 8, 32, 100, 1024):
        v = np.random.RandomState(3000 + n).normal(0, 1, n)
        out[f'daub4_{n}_in'] = v
        out[f'daub4_{n}_fwd'] = ms.dwt_daub4(v)
        out[f'daub4_{n}_inv'] = ms.dwt_daub4(v, True)
    # log_prior vectors
    post = rs.normal([3.0, 0.5], [0.5, 0.05], (40, 2))
    out['logprior_post'] = post
    out['logprior_value'] = ms.log_prior(
        post, np.array([3.0, 0.0, 0.5]), np.array([0.5, 0.0, 0.02]),
        np.array([0.25, 0.0, 0.04]), np.array([1.0, 0.0, 0.1]))
    # Gelman-Rubin (mc3/stats/gelman.py:12-92): a regular strided layout (what mrw / demc
    # / the GPU write), three burn-in values, and an irregular one (chains of unequal
    # length in arrival order)
    import mc3.stats.gelman as gel
    rs3 = np.random.RandomState(4040)
    nch, M0g, nit, npar = 6, 12, 90, 3
    drift = rs3.normal(0, 0.3, (nch, npar))
    Zg = np.concatenate([rs3.normal(0, 1, (M0g, npar)),
                         (rs3.normal(0, 1, (nit, nch, npar)) + drift).reshape(-1, npar)])
    zcg = np.concatenate([-np.ones(M0g, int), np.tile(np.arange(nch), nit)])
    out['gr_Z'] = Zg
    out['gr_zchain'] = zcg
    out['gr_burnins'] = np.array([0, 7, 40])
    out['gr_psrf'] = np.array([gel.gelman_rubin(Zg, zcg, int(b)) for b in out['gr_burnins']])
    order = rs3.permutation(nit*nch)[:400]
    order.sort()
    Zi = np.concatenate([Zg[:M0g], Zg[M0g:][order]])
    zci = np.concatenate([zcg[:M0g], zcg[M0g:][order]])
    out['gr_irregular_Z'] = Zi
    out['gr_irregular_zchain'] = zci
    out['gr_irregular_psrf'] = gel.gelman_rubin(Zi, zci, 5)
    # sample statistics (mc3/stats/stats.py:876-964, calc_hpd=True): a skewed 3-parameter
    # posterior inside a 5-parameter vector with one fixed and one shared parameter
    rs4 = np.random.RandomState(5050)
    post = np.column_stack([rs4.normal(3.0, 0.2, 30000), rs4.gamma(3.0, 0.5, 30000),
                            rs4.normal(-1.0, 2.0, 30000)**2*0.01 + rs4.normal(0, 0.05, 30000)])
    sbest = np.array([3.01, 7.0, 1.4, 0.02, 3.01])
    sstep = np.array([0.1, 0.0, 0.1, 0.1, -1.0])
    out['stats_posterior'] = post
    out['stats_bestp'] = sbest
    out['stats_pstep'] = sstep
    out['stats_result'] = np.array(ms.calc_sample_statistics(post, sbest, sstep, calc_hpd=True))
    np.savez_compressed(os.path.join(HERE, 'kernel_vectors.npz'), **out)
    print('kernel_vectors:', len(out), 'arrays')


if __name__ == '__main__' and not {'tutorial', 'residues', 'headline'} & set(sys.argv):
    kernel_vectors()
    chain_cases()


def tutorial_summary():
    """BASELINE.json configs[0]: examples/tutorial.py (quadratic fit, snooker,
    nchains=7, nsamples=1e5, burnin=1000) run by the reference itself; only summary
    statistics are kept (the comparison is statistical: different generator)."""
    np.random.seed(314)
    x = np.linspace(0, 10, 1000)
    p_true = [3.0, -2.4, 0.5]
    y = quad(p_true, x)
    uncert = np.sqrt(np.abs(y))
    data = y + np.random.normal(0, uncert)
    np.random.seed(2)
    cwd = os.getcwd()
    os.chdir('/tmp')
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out = mc3.sample(
                data=data, uncert=uncert, func=quad, params=np.array([10.0, -2.0, 0.1]),
                indparams=[x], pmin=np.array([-10.0, -20.0, -10.0]),
                pmax=np.array([40.0, 20.0, 10.0]), pstep=np.array([1.0, 0.5, 0.1]),
                sampler='snooker', nsamples=1e5, burnin=1000, nchains=7, ncpu=1,
                grtest=True, plots=False, log=None)
    finally:
        os.chdir(cwd)
    np.savez_compressed(
        os.path.join(HERE, 'tutorial_summary.npz'),
        meanp=out['meanp'], stdp=out['stdp'], medianp=out['medianp'],
        median_low_bounds=out['median_low_bounds'],
        median_high_bounds=out['median_high_bounds'], bestp=out['bestp'],
        best_chisq=out['best_chisq'], red_chisq=out['red_chisq'], BIC=out['BIC'],
        stddev_residuals=out['stddev_residuals'],
        acceptance_rate=out['acceptance_rate'], nrows=len(out['posterior']))
    print('tutorial_summary:', out['meanp'], out['stdp'], out['best_chisq'],
          out['acceptance_rate'])


if __name__ == '__main__' and 'tutorial' in sys.argv:
    tutorial_summary()


# ---- round 3: the three parity residues of VERDICT round 2 -------------------------------
class _HubRandom:
    """np.random as the hub (mc3/mcmc_driver.py:229-239) sees it: every trial vector of the
    kick-off is recorded in the order drawn."""
    def __init__(self):
        self.trials = []

    def normal(self, *a, **k):
        v = np.random.normal(*a, **k)
        self.trials.append(np.array(v, float))
        return v

    def uniform(self, *a, **k):
        v = np.random.uniform(*a, **k)
        self.trials.append(np.array(v, float))
        return v

    def __getattr__(self, name):
        return getattr(np.random, name)


class _HubProxy:
    def __init__(self):
        self.random = _HubRandom()

    def __getattr__(self, name):
        return getattr(np, name)


def kickoff_case(out, name, hub_seed, func, **kw):
    """The reference's own kick-off (mcmc_driver.py:229-270) on a short run: the tape of its
    trial vectors, the rows and log-posteriors it kept."""
    hub = _HubProxy()
    sys.modules['mc3.mcmc_driver'].np = hub
    np.random.seed(hub_seed)
    cwd = os.getcwd()
    os.chdir('/tmp')
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            mc3.sample(func=func, ncpu=1, plots=False, grtest=False, log=None, **kw)
    finally:
        os.chdir(cwd)
        sys.modules['mc3.mcmc_driver'].np = np
    npar = len(kw['params'])
    tape = np.array(hub.random.trials)
    M0 = kw.get('hsize', 10)*kw['nchains']
    assert len(_captured['Z0']) == M0
    for key, val in dict(
            tape=tape, Z0=_captured['Z0'], log_post0=_captured['log_post0'],
            data=kw['data'], uncert=kw['uncert'], x=kw['indparams'][0],
            params=kw['params'], pstep=kw['pstep'],
            pmin=kw.get('pmin', np.tile(-np.inf, npar)), pmax=kw.get('pmax', np.tile(np.inf, npar)),
            prior=kw.get('prior', np.zeros(npar)), priorlow=kw.get('priorlow', np.zeros(npar)),
            priorup=kw.get('priorup', np.zeros(npar)), nchains=kw['nchains'],
            hsize=kw.get('hsize', 10), wlike=kw.get('wlike', False),
            uniform=kw.get('kickoff', 'normal') == 'uniform').items():
        out[f'kick_{name}_{key}'] = np.asarray(val)
    print(f'kickoff {name}: {len(tape)} trials drawn for {M0} rows')


def residue_vectors():
    out = {}
    rs = np.random.RandomState(31415)
    # (a) log_prior with a two-sided Gaussian, a log-uniform (priorlow < 0) and a uniform
    # free parameter around a fixed one (mc3/stats/stats.py:287-392)
    post = np.column_stack([rs.normal(3.0, 0.5, 60), rs.uniform(0.05, 4.0, 60),
                            rs.normal(-1.0, 2.0, 60)])
    lp_prior = np.array([3.0, 0.0, 1.0, 0.0])
    lp_low = np.array([0.5, 0.0, -1.0, 0.0])
    lp_up = np.array([0.25, 0.0, 0.0, 0.0])
    lp_step = np.array([1.0, 0.0, 0.3, 2.0])
    out['logprior2_post'] = post
    out['logprior2_prior'], out['logprior2_priorlow'] = lp_prior, lp_low
    out['logprior2_priorup'], out['logprior2_pstep'] = lp_up, lp_step
    out['logprior2_value'] = ms.log_prior(post, lp_prior, lp_low, lp_up, lp_step)
    # (b) Gaussian chisq of BASELINE config 5's design-matrix model at its full size
    # (1e5 points x 10 columns): the model by NumPy, chisq by the reference's _chisq
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers as H
    prob = H.linear_problem(ndata=100000, nmodel=10, nchains=16, nsamples=16)
    A, data, uncert = prob['indep'], prob['data'], prob['uncert']
    P = prob['params'] + rs.normal(0, 0.03, (4, 10))
    out['c5_params'] = P
    out['c5_data_checksum'] = np.array([data.sum(), A.sum()])
    out['c5_chisq'] = np.array([ms.chisq(A.T @ p, data, uncert) for p in P])
    out['c5_chisq_priors'] = np.array([
        ms.chisq(A.T @ p, data, uncert, p, prob['prior'], prob['priorlow'], prob['priorup'])
        for p in P])
    # (c) the kick-off: the hub's trial vectors as drawn, and what it kept
    x, data, uncert = tutorial_data(100)
    base = dict(data=data, uncert=uncert, indparams=[x], sampler='mrw', burnin=0)
    kickoff_case(out, 'loose', 21, quad, nchains=7, nsamples=70,
                 params=np.array([10.0, -2.0, 0.1]), pstep=np.array([1.0, 0.5, 0.1]),
                 pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]), **base)
    kickoff_case(out, 'tight', 22, quad, nchains=6, nsamples=60, hsize=5,
                 params=np.array([4.0, -2.5, 0.5]), pstep=np.array([1.0, 0.5, 0.1]),
                 pmin=np.array([3.5, -2.8, 0.45]), pmax=np.array([6.5, -2.2, 0.56]), **base)
    x64, data64, uncert64 = tutorial_data(64)
    # a shared and a fixed parameter, uniform kick-off.  (A shared parameter whose own interval
    # is narrower than its source's cannot be recorded: once its copy lands outside, the
    # reference tests that stale copy forever, mcmc_driver.py:249-255, and gives up.)
    kickoff_case(out, 'shared_uniform', 23, line_shared, nchains=5, nsamples=50, hsize=8,
                 data=data64, uncert=uncert64, indparams=[x64], sampler='mrw', burnin=0,
                 params=np.array([3.2, -2.2, 0.45, -2.2, 1.0]),
                 pstep=np.array([0.3, 0.1, 0.03, -2, 0.0]),
                 pmin=np.array([2.0, -3.0, 0.42, -3.0, 0.0]),
                 pmax=np.array([4.5, -1.9, 0.56, -1.9, 2.0]), kickoff='uniform')
    kickoff_case(out, 'shared_normal', 24, line_shared, nchains=5, nsamples=50,
                 data=data64, uncert=uncert64, indparams=[x64], sampler='mrw', burnin=0,
                 params=np.array([3.2, -2.2, 0.45, -2.2, 1.0]),
                 pstep=np.array([0.3, 0.1, 0.03, -2, 0.0]),
                 pmin=np.array([2.0, -2.4, 0.42, -2.4, 0.0]),
                 pmax=np.array([4.5, -2.1, 0.56, -2.1, 2.0]),
                 prior=np.array([3.0, 0.0, 0.5, 0.0, 0.0]),
                 priorlow=np.array([0.5, 0.0, 0.02, 0.0, 0.0]),
                 priorup=np.array([0.25, 0.0, 0.04, 0.0, 0.0]))
    rs99 = np.random.RandomState(99)
    xw = np.linspace(0, 10, 64)
    dw = quad([3.0, -2.4, 0.5], xw) + rs99.normal(0, 1.0, 64) + 0.5*np.sin(2*np.pi*xw)
    kickoff_case(out, 'wavelet', 25, quad, nchains=7, nsamples=70, wlike=True,
                 data=dw, uncert=np.ones(64), indparams=[xw], sampler='mrw', burnin=0,
                 params=np.array([3.0, -2.4, 0.5, 1.0, 0.5, 1.0]),
                 pstep=np.array([0.3, 0.1, 0.02, 0.0, 0.1, 0.1]),
                 pmin=np.array([-10.0, -20.0, -10.0, 0.0, 0.2, 0.2]),
                 pmax=np.array([40.0, 20.0, 10.0, 2.0, 10.0, 10.0]))
    np.savez_compressed(os.path.join(HERE, 'residue_vectors.npz'), **out)
    print('residue_vectors:', len(out), 'arrays')


if __name__ == '__main__' and 'residues' in sys.argv:
    residue_vectors()


# ---- round 4: the lockstep semantics pinned to the reference at the headline shape -------
def _summary(out, burn_rows=0):
    post = np.asarray(out['posterior'])
    q = np.percentile(post, [2.5, 16, 50, 84, 97.5], axis=0)
    return dict(meanp=post.mean(0), stdp=post.std(0), quantiles=q,
                acceptance_rate=out['acceptance_rate'], best_chisq=out['best_chisq'],
                bestp=out['bestp'], nrows=len(post))


def headline_summary():
    """BASELINE.json configs[1] (256 chains, demc, ndata 1000) and a wavelet-likelihood run
    (64 chains, 2^10 points, snooker) by the reference itself, ncpu=1: posterior means, widths,
    quantiles, acceptance rate, best chisq; plus a few hundred mid-run rows with the chisq the
    reference's own _chisq / _dwt give for them (the direct 1e-10 leg at C2 / C4 sizes)."""
    save = {}
    x, data, uncert = tutorial_data(1000)
    np.random.seed(5)
    cwd = os.getcwd()
    os.chdir('/tmp')
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out = mc3.sample(
                data=data, uncert=uncert, func=quad, params=np.array([10.0, -2.0, 0.1]),
                indparams=[x], pmin=np.array([-10.0, -20.0, -10.0]),
                pmax=np.array([40.0, 20.0, 10.0]), pstep=np.array([1.0, 0.5, 0.1]),
                sampler='demc', nsamples=5e5, burnin=500, nchains=256, ncpu=1,
                grtest=False, plots=False, log=None)
    finally:
        os.chdir(cwd)
    for k, v in _summary(out).items():
        save['c2_' + k] = np.asarray(v)
    post = np.asarray(out['posterior'])
    rows = post[np.linspace(0, len(post) - 1, 300).astype(int)]
    save['c2_rows'] = rows
    save['c2_rows_chisq'] = np.array([ms.chisq(quad(p, x), data, uncert) for p in rows])
    print('c2:', save['c2_meanp'], save['c2_stdp'], save['c2_acceptance_rate'],
          save['c2_best_chisq'])

    # the wavelet likelihood: 64 chains, 2^10 points, snooker
    rs = np.random.RandomState(99)
    xw = np.linspace(0, 10, 1024)
    yw = quad([3.0, -2.4, 0.5], xw)
    dw = yw + rs.normal(0, 1.0, 1024) + 0.5*np.sin(2*np.pi*xw)
    wl = dict(data=dw, uncert=np.ones(1024), indparams=[xw],
              params=np.array([3.0, -2.4, 0.5, 1.0, 0.5, 1.0]),
              pstep=np.array([0.3, 0.1, 0.02, 0.0, 0.1, 0.1]),
              pmin=np.array([-10.0, -20.0, -10.0, 0.0, 0.01, 0.01]),
              pmax=np.array([40.0, 20.0, 10.0, 2.0, 10.0, 10.0]))
    np.random.seed(6)
    os.chdir('/tmp')
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out = mc3.sample(func=quad, sampler='snooker', nsamples=2e5, burnin=500,
                             nchains=64, ncpu=1, wlike=True, grtest=False, plots=False,
                             log=None, **wl)
    finally:
        os.chdir(cwd)
    for k, v in _summary(out).items():
        save['wl_' + k] = np.asarray(v)
    for k, v in wl.items():
        save['wl_in_' + k] = np.asarray(v[0] if k == 'indparams' else v)
    print('wl:', save['wl_meanp'], save['wl_stdp'], save['wl_acceptance_rate'],
          save['wl_best_chisq'])

    # C4-sized direct leg: 2^14 points, rows scattered around the wavelet posterior
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers as H
    prob = H.wavelet_problem(ndata=16384, nchains=8, nsamples=8)
    x4, d4 = prob['indep'], prob['data']
    rs2 = np.random.RandomState(77)
    P = np.tile(prob['params'], (200, 1))
    free = np.asarray(prob['pstep']) > 0
    P[:, free] += rs2.normal(0, 1.0, (200, free.sum()))*np.asarray(prob['pstep'])[free]*0.3
    P[:, 4:6] = np.abs(P[:, 4:6]) + 0.05
    save['c4_rows'] = P
    save['c4_rows_chisq'] = np.array([ms.dwt_chisq(quad(p[0:3], x4), d4, p) for p in P])
    save['c4_data_checksum'] = np.array([d4.sum(), x4.sum()])
    np.savez_compressed(os.path.join(HERE, 'headline_summary.npz'), **save)
    print('headline_summary:', len(save), 'arrays')


if __name__ == '__main__' and 'headline' in sys.argv:
    headline_summary()
