"""Randomised differential test: many small random problems (sampler, sizes, model
family, fixed/shared parameters, priors, bounds, thinning, epsilon, likelihood,
launch structure) -- the GPU must equal the lockstep oracle exactly on each."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def random_problem(seed):
    rs = np.random.RandomState(seed)
    wlike = rs.rand() < 0.25
    kind = 'poly' if (wlike or rs.rand() < 0.5) else 'linear'
    nmodel = int(rs.randint(1, 7)) if kind == 'poly' else int(rs.randint(1, 15))
    if wlike:
        ndata = int(2**rs.randint(1, 12))
    else:
        ndata = int(rs.choice([1, 2, 7, 64, 100, 1000, 1024, 1025, 3000, 20000]))
    nparams = nmodel + (3 if wlike else 0) + int(rs.randint(0, 3))
    sampler = str(rs.choice(['demc', 'snooker', 'mrw']))
    nchains = int(rs.randint(3, 40)) if ndata < 20000 else int(rs.randint(16, 40))
    thinning = int(rs.choice([1, 1, 2, 3]))
    x = np.sort(rs.uniform(-1, 1, ndata))
    if kind == 'poly':
        indep = x
        basis = np.array([x**k for k in range(nmodel)])
    else:
        indep = rs.normal(0, 1, (nmodel, ndata))
        basis = indep
    p_true = rs.normal(0, 1, nparams)
    if wlike:
        p_true[-3:] = [1.0, rs.uniform(0.2, 1.0), rs.uniform(0.5, 1.5)]
    data = p_true[:nmodel] @ basis + rs.normal(0, 1, ndata)
    uncert = rs.uniform(0.5, 2.0, ndata)
    pstep = np.full(nparams, 0.05)
    # a few fixed and shared parameters (never all)
    for j in range(nparams):
        r = rs.rand()
        if r < 0.15:
            pstep[j] = 0.0
        elif r < 0.3 and j > 0 and pstep[j - 1] > 0:
            src = int(rs.randint(0, j))
            if pstep[src] > 0:
                pstep[j] = -(src + 1)
                p_true[j] = p_true[src]
    if wlike:
        pstep[-3] = 0.0                      # gamma is fixed (only gamma = 1 is supported)
        pstep[-2:] = np.where(pstep[-2:] < 0, 0.05, pstep[-2:])
    if not np.any(pstep > 0):
        pstep[0] = 0.05
    lo = p_true - rs.uniform(0.05, 3.0, nparams)
    hi = p_true + rs.uniform(0.05, 3.0, nparams)
    for j in range(nparams):            # a shared parameter lives inside its source's bounds
        if pstep[j] < 0:
            src = -int(pstep[j]) - 1
            lo[j], hi[j] = lo[src], hi[src]
    if wlike:
        lo[-2:] = 0.01
    prior = np.zeros(nparams)
    plow = np.zeros(nparams)
    pup = np.zeros(nparams)
    for j in range(nparams):
        if pstep[j] > 0 and rs.rand() < 0.3:
            prior[j] = p_true[j] + rs.normal(0, 0.1)
            plow[j], pup[j] = rs.uniform(0.05, 1.0, 2)
    return dict(data=data, uncert=uncert, indep=indep, params=p_true, pstep=pstep,
                pmin=lo, pmax=hi, prior=prior, priorlow=plow, priorup=pup,
                nchains=nchains, nsamples=nchains*int(rs.randint(8, 30)),
                sampler=sampler, wlike=wlike, thinning=thinning,
                fgamma=float(rs.uniform(0.5, 1.5)),
                fepsilon=float(rs.choice([0.0, 0.0, 0.01])), hsize=int(rs.randint(1, 5)),
                model=kind, nmodel=nmodel, seed=int(rs.randint(0, 2**31)))


@pytest.mark.parametrize('seed', range(150))
def test_random_problem(seed):
    from mc3_amd import engine
    prob = random_problem(1000 + seed)
    rs = np.random.RandomState(seed)
    mode = rs.choice(['default', 'no_persist', 'eager'])
    kw = {'default': {}, 'no_persist': {'no_persist': True},
          'eager': {'no_persist': True, 'no_graph': True}}[mode]
    wpc = rs.choice(['', '1', '4', '16'])
    if wpc:
        os.environ['MC3HIP_WPC'] = wpc
    try:
        nz = int(np.ceil(prob['nsamples']/prob['nchains']/prob['thinning']))
        niter = nz*prob['thinning']
        gpu = engine.DeviceChains(trace_iters=niter, **prob, **kw)
    finally:
        os.environ.pop('MC3HIP_WPC', None)
    orc = H.make_oracle(prob, tile=gpu.plan['tile'])
    Z0, lp0 = H.initial_population(orc, seed=seed)
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(trace=True, nthreads=4)
    gpu.run()
    Z, lp, zc = gpu.rows()
    st = gpu.state()
    tc, tf = gpu.trace(niter)
    gpu.close()
    label = f"{prob['sampler']} {prob['model']}{prob['nmodel']} wlike={prob['wlike']} " \
            f"ndata={len(prob['data'])} nchains={prob['nchains']} thin={prob['thinning']} " \
            f"mode={mode} wpc={wpc or 'auto'}"
    np.testing.assert_array_equal(tf, orc.trace_flag, err_msg=label)
    np.testing.assert_array_equal(tc, orc.trace_chisq, err_msg=label)
    np.testing.assert_array_equal(Z, orc.Z, err_msg=label)
    np.testing.assert_array_equal(lp, orc.log_post, err_msg=label)
    np.testing.assert_array_equal(zc, orc.zchain, err_msg=label)
    np.testing.assert_array_equal(st['numaccept'], orc.numaccept, err_msg=label)
    np.testing.assert_array_equal(st['outbounds'], orc.outbounds, err_msg=label)
    np.testing.assert_array_equal(st['best_chisq'], orc.best_chisq, err_msg=label)
