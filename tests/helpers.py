"""Shared problem builders for the parity tests (CPU oracle and GPU)."""
import numpy as np

import ledger   # (a process that uses these builders leaves its launch ledger: tests/ledger.py)

ledger.install()



def quad(p, x):
    return p[0] + p[1]*x + p[2]*x**2.0


def tutorial_data(ndata, seed=314):
    """examples/tutorial.py:22-29 of the reference at a chosen size."""
    rs = np.random.RandomState(seed)
    x = np.linspace(0, 10, ndata)
    y = quad([3.0, -2.4, 0.5], x)
    uncert = np.sqrt(np.abs(y))
    data = y + rs.normal(0, uncert)
    return x, data, uncert


def quad_problem(ndata=100, nchains=8, nsamples=800, sampler='demc', **kw):
    x, data, uncert = tutorial_data(ndata)
    prob = dict(data=data, uncert=uncert, indep=x,
                params=np.array([10.0, -2.0, 0.1]),
                pstep=np.array([1.0, 0.5, 0.1]),
                pmin=np.array([-10.0, -20.0, -10.0]),
                pmax=np.array([40.0, 20.0, 10.0]),
                nchains=nchains, nsamples=nsamples, sampler=sampler,
                model='poly', seed=12345)
    prob.update(kw)
    return prob


def lean_shape_problem(nchains, nref, ndata=1000, sampler='demc', thinning=1, fx=0, **kw):
    """The shapes the hand-laid-out one-wave loops serve: the tutorial's quadratic (fx = 0), or
    a cubic with p2 fixed, p3 shared with p1 and two-sided Gaussian priors (fx = 1)."""
    if not fx:
        prob = quad_problem(ndata=ndata, nchains=nchains, nsamples=nchains*nref*thinning,
                            sampler=sampler, thinning=thinning)
    else:
        prob = shared_problem(nchains=nchains, nsamples=nchains*nref*thinning, sampler=sampler,
                              thinning=thinning)
        prob['params'] = prob['params'][:4]
        for k in ('pstep', 'pmin', 'pmax', 'prior', 'priorlow', 'priorup'):
            prob[k] = prob[k][:4]
        prob['pstep'] = np.array([0.3, 1.0, 0.0, -2.0])     # p2 fixed, p3 <- p1
    if sampler == 'mrw':
        prob['pstep'] = prob['pstep']*np.where(prob['pstep'] > 0, 0.05, 1.0)
    prob.update(kw)
    return prob


def shared_problem(nchains=8, nsamples=1600, sampler='snooker', **kw):
    """Degree-4 polynomial with tight bounds, two-sided Gaussian priors, a
    shared parameter (p3 <- p1) and a fixed one (p4)."""
    x, data, uncert = tutorial_data(64)
    x = x/10.0
    prob = dict(data=data, uncert=uncert, indep=x,
                params=np.array([3.2, -22.0, 45.0, -22.0, 1.0]),
                pstep=np.array([0.3, 1.0, 3.0, -2, 0.0]),
                pmin=np.array([2.0, -30.0, 30.0, -30.0, 0.0]),
                pmax=np.array([4.5, -15.0, 60.0, -15.0, 2.0]),
                prior=np.array([3.0, 0.0, 50.0, 0.0, 0.0]),
                priorlow=np.array([0.5, 0.0, 2.0, 0.0, 0.0]),
                priorup=np.array([0.25, 0.0, 4.0, 0.0, 0.0]),
                nchains=nchains, nsamples=nsamples, sampler=sampler,
                model='poly', seed=777)
    prob.update(kw)
    return prob


def wavelet_problem(ndata=64, nchains=7, nsamples=700, sampler='snooker', **kw):
    rs = np.random.RandomState(99)
    x = np.linspace(0, 10, ndata)
    data = quad([3.0, -2.4, 0.5], x) + rs.normal(0, 1.0, ndata) \
        + 0.5*np.sin(2*np.pi*x)
    prob = dict(data=data, uncert=np.ones(ndata), indep=x,
                params=np.array([3.0, -2.4, 0.5, 1.0, 0.5, 1.0]),
                pstep=np.array([0.3, 0.1, 0.02, 0.0, 0.1, 0.1]),
                pmin=np.array([-10.0, -20.0, -10.0, 0.0, 0.01, 0.01]),
                pmax=np.array([40.0, 20.0, 10.0, 2.0, 10.0, 10.0]),
                nchains=nchains, nsamples=nsamples, sampler=sampler,
                wlike=True, model='poly', seed=4321)
    prob.update(kw)
    return prob


def linear_problem(ndata=512, nmodel=10, nchains=16, nsamples=1600,
                   sampler='demc', **kw):
    """C5-shaped: Legendre/cosine design matrix, Gaussian priors on three
    parameters, bounds on all."""
    rs = np.random.RandomState(7)
    t = np.linspace(-1, 1, ndata)
    cols = []
    for k in range(nmodel):
        if k % 2 == 0:
            cols.append(np.polynomial.legendre.legval(t, [0]*(k//2) + [1]))
        else:
            cols.append(np.cos(np.pi*(k//2 + 1)*t))
    A = np.array(cols)
    p_true = rs.uniform(-1, 1, nmodel)
    # (accumulated term by term: a BLAS product may split the sum differently with the number
    # of threads it is given -- torchrun's workers get one -- and every process of a
    # multi-rank test must build the same data to the last bit)
    signal = np.zeros(ndata)
    for k in range(nmodel):
        signal = signal + p_true[k]*A[k]
    data = signal + rs.normal(0, 1, ndata)
    prior = np.zeros(nmodel)
    plow = np.zeros(nmodel)
    pup = np.zeros(nmodel)
    prior[:3] = p_true[:3]
    plow[:3] = 1.0
    pup[:3] = 1.0
    prob = dict(data=data, uncert=np.ones(ndata), indep=A,
                params=p_true + rs.normal(0, 0.05, nmodel),
                pstep=np.full(nmodel, 0.05),
                pmin=np.full(nmodel, -5.0), pmax=np.full(nmodel, 5.0),
                prior=prior, priorlow=plow, priorup=pup,
                nchains=nchains, nsamples=nsamples, sampler=sampler,
                model='linear', seed=2024)
    prob.update(kw)
    return prob


def wavelet_linear_problem(ndata=64, nmodel=4, nchains=7, nsamples=7*30, sampler='snooker',
                           **kw):
    """Wavelet likelihood with a design-matrix model (Legendre columns): the model call of
    chain.py:317-319 feeding src_c/_dwt.c:76-114."""
    rs = np.random.RandomState(5)
    t = np.linspace(-1, 1, ndata)
    A = np.array([np.polynomial.legendre.legval(t, [0]*k + [1]) for k in range(nmodel)])
    p_true = rs.uniform(-1, 1, nmodel)
    signal = np.zeros(ndata)
    for k in range(nmodel):
        signal = signal + p_true[k]*A[k]
    data = signal + rs.normal(0, 1, ndata) + 0.5*np.sin(6*np.pi*t)
    prob = dict(data=data, uncert=np.ones(ndata), indep=A,
                params=np.concatenate([p_true, [1.0, 0.5, 1.0]]),
                pstep=np.concatenate([np.full(nmodel, 0.05), [0.0, 0.1, 0.1]]),
                pmin=np.concatenate([np.full(nmodel, -5.0), [0.0, 0.01, 0.01]]),
                pmax=np.concatenate([np.full(nmodel, 5.0), [2.0, 10.0, 10.0]]),
                nchains=nchains, nsamples=nsamples, sampler=sampler, wlike=True,
                model='linear', seed=77)
    prob.update(kw)
    return prob


def make_oracle(prob, plan=None, **extra):
    """The lockstep oracle for a problem; ``plan`` (DeviceChains.plan) makes it sum the
    data in the order the library reports: tile size, virtual lanes and -- wavelet
    likelihood -- the threads that share a chain."""
    kw = dict(prob)
    if plan is not None:
        kw.update(tile=plan['tile'], vlanes=plan['vlanes'],
                  tile_tree=plan.get('tile_tree', 0))
        if prob.get('wlike'):
            kw['wl_threads'] = 64*plan['waves_per_chain']
    kw.update(extra)
    from oracle import lockstep as ls      # (lazily: tools/ import this module for the
    return ls.LockstepRun(**kw)            # problem builders only and must not load oracle/)


def initial_population(orc, seed=5):
    """M0 in-bounds draws around the initial guess with finite chisq
    (mcmc_driver.py:229-262), evaluated by the oracle so that every
    implementation under test starts from the same rows."""
    rs = np.random.RandomState(seed)
    ifree = orc.ifree
    x0 = orc.params[ifree]
    sig = orc.pstep[ifree]
    Z0 = np.empty((orc.M0, orc.nfree))
    lp0 = np.empty(orc.M0)
    values = orc.params.copy()
    filled = 0
    failed = 0
    while filled < orc.M0:
        failed += 1
        if failed > 200*orc.M0 + 1000:
            raise ValueError('Cannot populate an initial sample set of parameters')
        trial = rs.normal(x0, sig)
        values[ifree] = trial
        if np.any(values > orc.pmax) or np.any(values < orc.pmin):
            continue
        for s, src in zip(orc.ishare, orc.share_src):
            values[s] = values[src]
        c = orc.eval_chisq(values)[0]
        if not np.isfinite(c):
            continue
        Z0[filled] = trial
        lp0[filled] = -0.5*c
        filled += 1
    return Z0, lp0


def gpu_count():
    """Devices HIP shows, counted in a CHILD process (the caller's own GPU state is left alone);
    0 when there is no runtime or no GPU.  (MC3_TEST_FAKE_GPUS=n: the rehearsal of the node's
    tests on a box without one, tests/test_multidevice_dry_run.py.)"""
    import os
    import subprocess
    import sys
    if os.environ.get('MC3_TEST_FAKE_GPUS'):
        return int(os.environ['MC3_TEST_FAKE_GPUS'])
    code = ('import ctypes\n'
            'n = ctypes.c_int(0)\n'
            'rc = ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n))\n'
            'print(n.value if rc == 0 else 0)\n')
    try:
        res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True,
                             timeout=120)
        return int(res.stdout.strip() or 0)
    except (OSError, ValueError, subprocess.SubprocessError):
        return 0
