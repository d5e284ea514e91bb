"""Randomised differential test: many small random problems (sampler, sizes, model
family, fixed/shared parameters, priors, bounds, thinning, epsilon, likelihood,
launch structure) -- the GPU must equal the lockstep oracle exactly on each.

Two families: ``random_problem`` spans everything the library accepts (run through the
general kernels with the per-step trace switched on for every other seed, and without
it -- i.e. with the lean kernels allowed -- for the rest); ``lean_problem`` draws from
the shapes the lean kernels serve (lean_demc_kernel and persist_fast_kernel with their
FX / MULTI / TH instances: the kernels that carry the headline number) and asserts that
the library really chose them."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def random_problem(seed, wavelet_any_model=False):
    """wavelet_any_model: the wavelet likelihood with either model kind and up to 2^14 points
    (the nine-wave kernel's sizes included); the default sequence draws polynomials only under
    the wavelet likelihood and is kept as it is (its cases are seeded)."""
    rs = np.random.RandomState(seed)
    wlike = rs.rand() < 0.25
    kind = 'poly' if (wlike or rs.rand() < 0.5) else 'linear'
    if wavelet_any_model:
        wlike = True
        kind = str(rs.choice(['poly', 'linear', 'linear']))
    nmodel = int(rs.randint(1, 7)) if kind == 'poly' else int(rs.randint(1, 15))
    if wavelet_any_model:
        ndata = int(2**rs.choice([1, 3, 6, 9, 11, 12, 13, 13, 14, 14]))
    elif wlike:
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


def lean_problem(seed):
    """A problem the lean kernels serve: polynomial with 2..4 coefficients and nothing
    but coefficients among the parameters, Gaussian chisq, no support distribution."""
    rs = np.random.RandomState(seed)
    nmodel = int(rs.randint(2, 5))
    flavour = str(rs.choice(['plain', 'plain', 'fx', 'multi', 'thin']))
    if flavour == 'multi':
        ndata = int(rs.choice([1025, 1500, 2048, 2049, 3000, 4096]))
    else:
        ndata = int(rs.choice([1, 3, 64, 100, 777, 1000, 1024]))
    sampler = str(rs.choice(['demc', 'demc', 'snooker', 'mrw']))
    nchains = int(rs.randint(3, 48))
    thinning = int(rs.choice([2, 3, 5])) if flavour == 'thin' else 1
    x = np.sort(rs.uniform(-1, 1, ndata))
    p_true = rs.normal(0, 1, nmodel)
    data = p_true @ np.array([x**k for k in range(nmodel)]) + rs.normal(0, 1, ndata)
    uncert = rs.uniform(0.5, 2.0, ndata)
    pstep = np.full(nmodel, 0.05)
    prior = np.zeros(nmodel)
    plow = np.zeros(nmodel)
    pup = np.zeros(nmodel)
    if flavour == 'fx':
        # fixed / shared coefficients and Gaussian priors, in any mixture (never all fixed)
        for j in range(nmodel):
            r = rs.rand()
            if r < 0.3:
                pstep[j] = 0.0
            elif r < 0.5 and j > 0 and pstep[0] > 0:
                pstep[j] = -1.0
                p_true[j] = p_true[0]
        if not np.any(pstep > 0):
            pstep[0] = 0.05
        for j in range(nmodel):
            if pstep[j] > 0 and rs.rand() < 0.5:
                prior[j] = p_true[j] + rs.normal(0, 0.1)
                plow[j], pup[j] = rs.uniform(0.05, 1.0, 2)
    lo = p_true - rs.uniform(0.05, 3.0, nmodel)
    hi = p_true + rs.uniform(0.05, 3.0, nmodel)
    for j in range(nmodel):
        if pstep[j] < 0:
            lo[j], hi[j] = lo[0], hi[0]
    return dict(data=data, uncert=uncert, indep=x, params=p_true, pstep=pstep,
                pmin=lo, pmax=hi, prior=prior, priorlow=plow, priorup=pup,
                nchains=nchains, nsamples=nchains*thinning*int(rs.randint(20, 200)),
                sampler=sampler, wlike=False, thinning=thinning,
                fgamma=float(rs.uniform(0.5, 1.5)), fepsilon=0.0,
                hsize=int(rs.randint(1, 5)), model='poly', nmodel=nmodel,
                seed=int(rs.randint(0, 2**31))), flavour


def compare_final(gpu, orc, label):
    Z, lp, zc = gpu.rows()
    st = gpu.state()
    np.testing.assert_array_equal(Z, orc.Z, err_msg=label)
    np.testing.assert_array_equal(lp, orc.log_post, err_msg=label)
    np.testing.assert_array_equal(zc, orc.zchain, err_msg=label)
    np.testing.assert_array_equal(st['numaccept'], orc.numaccept, err_msg=label)
    np.testing.assert_array_equal(st['outbounds'], orc.outbounds, err_msg=label)
    np.testing.assert_array_equal(st['best_chisq'], orc.best_chisq, err_msg=label)
    np.testing.assert_array_equal(st['best_iter'], orc.best_iter, err_msg=label)
    np.testing.assert_array_equal(st['bestx'][st['best_iter'] >= 0],
                                  orc.bestx[orc.best_iter >= 0], err_msg=label)
    np.testing.assert_array_equal(st['chisq'], orc.chisq, err_msg=label)
    np.testing.assert_array_equal(st['X'], orc.X, err_msg=label)


@pytest.mark.parametrize('seed', range(120))
def test_lean_kernels_random_problem(seed):
    """The kernels behind the headline number, fuzzed: the library must pick a lean
    instance for these shapes (plan.fast) and equal the oracle exactly."""
    from mc3_amd import engine
    prob, flavour = lean_problem(5000 + seed)
    gpu = engine.DeviceChains(**prob)
    plan = gpu.plan
    label = f"{flavour} {prob['sampler']} poly{prob['nmodel']} ndata={len(prob['data'])} " \
            f"nchains={prob['nchains']} thin={prob['thinning']} pstep={prob['pstep']} plan={plan}"
    assert plan['persistent'] > 0 and plan['fast'] > 0, label
    # (hand-laid-out loops: thinning 1 with one, two or streamed tiles; any thinning as long as
    # the data are one tile)
    lean_loop = prob['thinning'] == 1 or len(prob['data']) <= 1024
    assert (plan['fast'] == 2) == lean_loop, label
    orc = H.make_oracle(prob, plan=plan)
    Z0, lp0 = H.initial_population(orc, seed=seed)
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(nthreads=4)
    # in two uneven pieces: the second launch resumes from device state
    nz = gpu.nzchain
    gpu.run(nz//3)
    gpu.run()
    assert gpu.plan['fast'] == plan['fast'], 'the persistent launch gave up: ' + label
    compare_final(gpu, orc, label)
    gpu.close()


@pytest.mark.parametrize('seed', range(150))
def test_random_problem(seed):
    from mc3_amd import engine
    prob = random_problem(1000 + seed)
    rs = np.random.RandomState(seed)
    mode = rs.choice(['default', 'no_persist', 'eager'])
    kw = {'default': {}, 'no_persist': {'no_persist': True},
          'eager': {'no_persist': True, 'no_graph': True}}[mode]
    wpc = rs.choice(['', '1', '4', '4'])      # (four draws: the seeds keep their problems)
    traced = seed % 2 == 0          # odd seeds: no per-step trace, lean kernels allowed
    if wpc:
        os.environ['MC3HIP_WPC'] = wpc
    try:
        nz = int(np.ceil(prob['nsamples']/prob['nchains']/prob['thinning']))
        niter = nz*prob['thinning']
        gpu = engine.DeviceChains(trace_iters=niter if traced else 0, **prob, **kw)
    finally:
        os.environ.pop('MC3HIP_WPC', None)
    orc = H.make_oracle(prob, plan=gpu.plan)
    Z0, lp0 = H.initial_population(orc, seed=seed)
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(trace=True, nthreads=4)
    gpu.run()
    Z, lp, zc = gpu.rows()
    st = gpu.state()
    label = f"{prob['sampler']} {prob['model']}{prob['nmodel']} wlike={prob['wlike']} " \
            f"ndata={len(prob['data'])} nchains={prob['nchains']} thin={prob['thinning']} " \
            f"mode={mode} wpc={wpc or 'auto'} traced={traced} plan={gpu.plan}"
    if traced:
        tc, tf = gpu.trace(niter)
        np.testing.assert_array_equal(tf, orc.trace_flag, err_msg=label)
        np.testing.assert_array_equal(tc, orc.trace_chisq, err_msg=label)
    gpu.close()
    np.testing.assert_array_equal(Z, orc.Z, err_msg=label)
    np.testing.assert_array_equal(lp, orc.log_post, err_msg=label)
    np.testing.assert_array_equal(zc, orc.zchain, err_msg=label)
    np.testing.assert_array_equal(st['numaccept'], orc.numaccept, err_msg=label)
    np.testing.assert_array_equal(st['outbounds'], orc.outbounds, err_msg=label)
    np.testing.assert_array_equal(st['best_chisq'], orc.best_chisq, err_msg=label)


@pytest.mark.parametrize('seed', range(40))
def test_random_wavelet_problem_either_model_kind(seed):
    """The wavelet likelihood with polynomials and design matrices up to 2^14 points, fixed and
    shared parameters, priors, thinning, epsilon; even seeds traced (the general chain), odd
    seeds not (the hand-structured chain wave where it applies; run in two pieces)."""
    from mc3_amd import engine
    prob = random_problem(9000 + seed, wavelet_any_model=True)
    traced = seed % 2 == 0
    nz = int(np.ceil(prob['nsamples']/prob['nchains']/prob['thinning']))
    niter = nz*prob['thinning']
    gpu = engine.DeviceChains(trace_iters=niter if traced else 0, **prob)
    orc = H.make_oracle(prob, plan=gpu.plan)
    try:
        Z0, lp0 = H.initial_population(orc, seed=seed)
    except ValueError:          # (bounds that leave no room around the start: not a case)
        gpu.close()
        pytest.skip('no initial population for this draw')
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(trace=True, nthreads=4)
    if not traced:              # (the trace buffer holds the steps of ONE run)
        gpu.run(gpu.nzchain//3)
    gpu.run()
    label = f"{prob['sampler']} {prob['model']}{prob['nmodel']} ndata={len(prob['data'])} " \
            f"nchains={prob['nchains']} thin={prob['thinning']} traced={traced} plan={gpu.plan}"
    if traced:
        tc, tf = gpu.trace(niter)
        np.testing.assert_array_equal(tf, orc.trace_flag, err_msg=label)
        np.testing.assert_array_equal(tc, orc.trace_chisq, err_msg=label)
    compare_final(gpu, orc, label)
    gpu.close()


# ---- round 4: the declared cadence, the EPS loop, FX x thinning, the run-time specialisations --
def round4_problem(seed):
    """A problem from the shapes round 4 put on hand-laid-out loops: zlag >= 1 (any sampler),
    fixed / shared coefficients and priors with thinning > 1 (demc), the support distribution
    (demc, strict lockstep), polynomials of 5..8 coefficients and design matrices of 1..10
    columns (compiled for the model when the context is created), ndata <= 1024."""
    rs = np.random.RandomState(seed)
    kind = str(rs.choice(['poly', 'polybig', 'linear', 'linear']))
    if kind == 'poly':
        nmodel, model = int(rs.randint(2, 5)), 'poly'
    elif kind == 'polybig':
        nmodel, model = int(rs.randint(5, 9)), 'poly'
    else:
        nmodel, model = int(rs.randint(1, 11)), 'linear'
    sampler = str(rs.choice(['demc', 'demc', 'snooker', 'mrw']))
    if sampler == 'mrw' and nmodel > 4:
        sampler = 'demc'                       # (the random-walk loop draws four deviates)
    ndata = int(rs.choice([1, 3, 64, 100, 777, 1000, 1024]))
    nchains = int(rs.randint(3, 48))
    fx = rs.rand() < 0.5
    thinning = int(rs.choice([1, 1, 2, 3, 5]))
    zlag = int(rs.choice([0, 1, 2, 3, 5]))
    feps = 0.0
    if sampler == 'demc' and thinning == 1 and zlag == 0 and rs.rand() < 0.7:
        feps = float(rs.choice([0.01, 0.1]))
    if sampler != 'demc' and fx and thinning > 1:
        thinning = 1                           # (FX x THIN exists for demc only)
    if sampler == 'snooker' and thinning == 1 and nmodel <= 4 and rs.rand() < 0.5:
        feps = float(rs.choice([0.01, 0.1]))   # (the snooker loop's EPS flavour: <= 4 deviates)
    x = np.sort(rs.uniform(-1, 1, ndata))
    if model == 'poly':
        indep = x
        basis = np.array([x**k for k in range(nmodel)])
    else:
        indep = rs.normal(0, 1, (nmodel, ndata))
        basis = indep
    p_true = rs.normal(0, 1, nmodel)
    data = p_true @ basis + rs.normal(0, 1, ndata)
    uncert = rs.uniform(0.5, 2.0, ndata)
    pstep = np.full(nmodel, 0.05)
    prior, plow, pup = np.zeros(nmodel), np.zeros(nmodel), np.zeros(nmodel)
    if fx:
        for j in range(nmodel):
            r = rs.rand()
            if r < 0.25:
                pstep[j] = 0.0
            elif r < 0.45 and j > 0 and pstep[0] > 0:
                pstep[j] = -1.0
                p_true[j] = p_true[0]
        if not np.any(pstep > 0):
            pstep[0] = 0.05
        for j in range(nmodel):
            if rs.rand() < 0.4:                # (also on fixed parameters: a constant term)
                prior[j] = p_true[j] + rs.normal(0, 0.1)
                plow[j], pup[j] = rs.uniform(0.05, 1.0, 2)
    if sampler == 'mrw' and np.sum(pstep > 0) > 4:
        sampler = 'demc'
    lo = p_true - rs.uniform(0.05, 3.0, nmodel)
    hi = p_true + rs.uniform(0.05, 3.0, nmodel)
    for j in range(nmodel):
        if pstep[j] < 0:
            lo[j], hi[j] = lo[0], hi[0]
    prob = dict(data=data, uncert=uncert, indep=indep, params=p_true, pstep=pstep,
                pmin=lo, pmax=hi, prior=prior, priorlow=plow, priorup=pup,
                nchains=nchains, nsamples=nchains*thinning*int(rs.randint(20, 120)),
                sampler=sampler, wlike=False, thinning=thinning,
                fgamma=float(rs.uniform(0.5, 1.5)), fepsilon=feps,
                hsize=int(rs.randint(1, 5)), model=model, nmodel=nmodel, zlag=zlag,
                seed=int(rs.randint(0, 2**31)))
    # does a hand-laid-out loop serve it?
    plain = not fx or (np.all(pstep > 0) and not np.any((plow > 0) & (pup > 0)))
    if sampler == 'demc':                      # (the EPS flavour draws four deviates)
        lean = feps == 0.0 or (thinning == 1 and zlag == 0 and np.sum(pstep > 0) <= 4)
    elif sampler == 'snooker':
        lean = (feps == 0.0 and (thinning == 1 or plain)) or (feps != 0.0 and thinning == 1)
    else:
        lean = thinning == 1 or plain
    jit = kind != 'poly'
    return prob, lean, jit, f'{kind}{nmodel} {sampler} thin={thinning} zlag={zlag} eps={feps} ' \
                            f'fx={fx} ndata={ndata} nchains={nchains} pstep={pstep}'


@pytest.mark.parametrize('seed', range(100))
def test_round4_kernels_random_problem(seed):
    from mc3_amd import engine
    prob, lean, jit, label = round4_problem(9000 + seed)
    gpu = engine.DeviceChains(**prob)
    plan = gpu.plan
    label += f' plan={plan} served={gpu.jit_served!r}'
    assert plan['persistent'] > 0 and plan['zlag'] == prob['zlag'], label
    if lean:
        assert plan['fast'] == 2 and plan['jit'] == int(jit), label
    orc = H.make_oracle(prob, plan=plan)
    Z0, lp0 = H.initial_population(orc, seed=seed)
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(nthreads=4)
    nz = gpu.nzchain
    gpu.run(nz//3)
    gpu.run()
    assert gpu.plan['fast'] == plan['fast'], 'the persistent launch gave up: ' + label
    compare_final(gpu, orc, label)
    gpu.close()


@pytest.mark.parametrize('seed', range(150))
def test_rows_kernels_random_problem(seed, monkeypatch):
    """kernels_rows.h, fuzzed: one-tile polynomial shapes (fixed / shared coefficients and priors
    among them) with two to four chains per wave
    (MC3HIP_PERSIST_CAPACITY pretends that fewer workgroups fit), every sampler, ragged
    ensembles, tight bounds, tiny histories; a third of the seeds behind MC3HIP_NO_ROWS (the general
    lean instance takes turns) -- both must equal the oracle exactly."""
    from mc3_amd import engine
    rs = np.random.RandomState(12000 + seed)
    prob = None
    for attempt in range(100):                 # (a one-tile, unthinned draw of lean_problem)
        cand, flavour = lean_problem(12000 + 97*seed + attempt)
        if flavour in ('plain', 'fx') and len(cand['data']) >= 3:
            prob = cand
            break
    assert prob is not None
    turns = int(rs.randint(2, 5))
    nchains = int(rs.randint(2*turns + 1, 60))
    capacity = -(-nchains//turns)
    if -(-nchains//capacity) != turns:         # (e.g. 9 chains over 3 turns of 3: two would do)
        nchains = capacity*turns - int(rs.randint(0, 2))
    prob.update(nchains=nchains, nsamples=nchains*int(rs.randint(20, 120)),
                sampler=str(rs.choice(['demc', 'snooker', 'mrw'])))
    if rs.rand() < 0.3 and flavour == 'plain':  # bounds that reject many proposals
        prob['pmin'] = prob['params'] - rs.uniform(0.02, 0.2, len(prob['params']))
        prob['pmax'] = prob['params'] + rs.uniform(0.02, 0.2, len(prob['params']))
    monkeypatch.setenv('MC3HIP_PERSIST_CAPACITY', str(capacity))
    layout = 'rows'
    gpu = engine.DeviceChains(**prob)
    plan = gpu.plan
    label = f"{layout} {flavour} {prob['sampler']} poly{prob['nmodel']} ndata={len(prob['data'])} " \
            f"nchains={nchains} turns={turns} hsize={prob['hsize']} pstep={prob['pstep']} plan={plan}"
    assert plan['persistent'] > 0 and plan['fast'] == (2 if layout == 'rows' else 1), label
    assert plan['chains_per_workgroup'] == turns, label
    orc = H.make_oracle(prob, plan=plan)
    Z0, lp0 = H.initial_population(orc, seed=seed)
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(nthreads=4)
    nz = gpu.nzchain
    gpu.run(nz//3)
    gpu.run()
    assert gpu.plan['fast'] == plan['fast'] and gpu.plan['recoveries'] == 0, label
    compare_final(gpu, orc, label)
    gpu.close()


def wide_problem(seed):
    """1 025 ... 16 384 points of the plain strict DEMC / snooker shapes with few chains: the
    tiles in the registers of four / eight waves per chain (lean_*_wide_kernel)."""
    rs = np.random.RandomState(640000 + seed)
    nmodel = int(rs.randint(2, 5))
    ndata = int(rs.choice([int(rs.randint(1025, 16385)), 1025, 2048, 2049, 4096, 4097, 8192, 8193,
                           16383, 16384]))
    sampler = str(rs.choice(['demc', 'snooker']))
    nchains = int(rs.randint(3, 24))
    x = rs.uniform(-1, 1, ndata)                       # (unsorted: any abscissae)
    p_true = rs.normal(0, 1, nmodel)
    data = p_true @ np.array([x**k for k in range(nmodel)]) + rs.normal(0, 1, ndata)
    uncert = rs.uniform(0.5, 2.0, ndata)
    lo = p_true - rs.uniform(0.02, 3.0, nmodel)
    hi = p_true + rs.uniform(0.02, 3.0, nmodel)
    return dict(data=data, uncert=uncert, indep=x, params=p_true, pstep=np.full(nmodel, 0.02),
                pmin=lo, pmax=hi, nchains=nchains, nsamples=nchains*int(rs.randint(20, 90)),
                sampler=sampler, wlike=False, thinning=1, fgamma=float(rs.uniform(0.5, 1.5)),
                fepsilon=0.0, hsize=int(rs.randint(1, 5)), model='poly', nmodel=nmodel,
                seed=int(rs.randint(0, 2**31)))


@pytest.mark.parametrize('seed', range(60))
def test_wide_kernels_random_problem(seed):
    """The round-5 loops with the data in the registers of several waves, fuzzed: the library
    must pick them (fast == 2, four or eight waves per chain) and equal the oracle exactly --
    across two launches."""
    from mc3_amd import engine
    prob = wide_problem(seed)
    gpu = engine.DeviceChains(**prob)
    plan = gpu.plan
    nd = len(prob['data'])
    label = f"{prob['sampler']} poly{prob['nmodel']} ndata={nd} nchains={prob['nchains']} plan={plan}"
    assert plan['persistent'] > 0 and plan['fast'] == 2 and plan['tile'] == 0, label
    assert plan['waves_per_chain'] == (8 if nd > 8192 else 4), label
    orc = H.make_oracle(prob, plan=plan)
    Z0, lp0 = H.initial_population(orc, seed=seed)
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(nthreads=4)
    gpu.run(gpu.nzchain//3)
    gpu.run()
    assert gpu.plan['fast'] == 2 and gpu.plan['recoveries'] == 0, label
    compare_final(gpu, orc, label)
    gpu.close()


def resident_async_problem(seed):
    """Design matrices of 2..11 columns on 16 384 ... 70 000 points, DEMC, thinning 1: the
    resident tiled plan with the owners as waves of their own and the blocks of sixteen chains
    made up at run time (resident_async_kernel) -- ragged columns of chains, ragged last tiles,
    fixed coefficients, priors, the support distribution, tight bounds."""
    rs = np.random.RandomState(650000 + seed)
    nmodel = int(rs.randint(2, 12))
    ndata = int(rs.choice([int(rs.randint(16384, 70000)), 16384, 16385, 20480, 32768, 40961]))
    nchains = int(rs.choice([int(rs.randint(16, 200)), 16, 17, 33, 48, 64, 129]))
    t = np.sort(rs.uniform(-1, 1, ndata))
    A = np.array([np.cos(0.7*k*t + 0.3*k) for k in range(nmodel)])
    p_true = rs.uniform(-1, 1, nmodel)
    data = p_true @ A + rs.normal(0, 1, ndata)
    uncert = rs.uniform(0.5, 2.0, ndata)
    pstep = np.full(nmodel, 0.03)
    params = p_true + rs.normal(0, 0.02, nmodel)
    if nmodel > 3 and rs.rand() < 0.4:
        pstep[int(rs.randint(1, nmodel))] = 0.0                  # a fixed coefficient
    width = rs.uniform(0.05, 3.0) if rs.rand() < 0.3 else 5.0    # tight bounds: many rejected
    prior = np.zeros(nmodel)
    plow = np.zeros(nmodel)
    pup = np.zeros(nmodel)
    if rs.rand() < 0.5:
        k = int(rs.randint(0, nmodel))
        if pstep[k] > 0:
            prior[k], plow[k], pup[k] = p_true[k], rs.uniform(0.2, 2), rs.uniform(0.2, 2)
    return dict(data=data, uncert=uncert, indep=A, params=params, pstep=pstep,
                pmin=params - width, pmax=params + width, prior=prior, priorlow=plow,
                priorup=pup, nchains=nchains, nsamples=nchains*int(rs.randint(8, 40)),
                sampler='demc', wlike=False, thinning=1, fgamma=float(rs.uniform(0.5, 1.5)),
                fepsilon=float(rs.choice([0.0, 0.0, 0.01])), hsize=int(rs.randint(1, 4)),
                model='linear', seed=int(rs.randint(0, 2**31)))


@pytest.mark.parametrize('seed', range(40))
def test_resident_async_random_problem(seed, monkeypatch):
    """resident_async_kernel fuzzed (which chains share a block, and when it is multiplied, is
    decided at run time by what has been proposed: every run takes another path through the
    same arithmetic) -- the oracle's rows, counters and best fits exactly, across two
    launches."""
    from mc3_amd import engine
    prob = resident_async_problem(seed)
    monkeypatch.setenv('MC3HIP_RS_ASYNC', '1')      # (whatever the size: decide_persist)
    gpu = engine.DeviceChains(**prob)
    plan = gpu.plan
    label = (f"linear{len(prob['params'])} ndata={len(prob['data'])} nchains={prob['nchains']} "
             f"plan={plan}")
    assert plan['resident'] == 1 and plan['fast'] == 4 and plan['tile'] == 2048, label
    orc = H.make_oracle(prob, plan=plan)
    try:
        Z0, lp0 = H.initial_population(orc, seed=seed)
    except ValueError:
        gpu.close()
        pytest.skip('no initial population within these bounds')
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(nthreads=4)
    gpu.run(max(1, gpu.nzchain//3))
    gpu.run()
    compare_final(gpu, orc, label)
    assert gpu.plan['recoveries'] == 0, label
    gpu.close()


def wide_offlean_problem(seed):
    """1 025 ... 8 192 points with what the library's wide loops do not carry -- fixed / shared
    coefficients, priors, the support distribution, the random walk: lean_*_body<.., NW = 4, NT>
    specialised at run time (ctx.h wide_jit_tiles)."""
    rs = np.random.RandomState(660000 + seed)
    prob = wide_problem(seed)
    nmodel = len(prob['params'])
    nd = int(rs.choice([int(rs.randint(1025, 8193)), 1025, 2048, 2049, 4096, 4097, 8192]))
    x = rs.uniform(-1, 1, nd)
    p_true = prob['params']
    prob.update(indep=x, uncert=rs.uniform(0.5, 2.0, nd),
                data=p_true @ np.array([x**k for k in range(nmodel)]) + rs.normal(0, 1, nd),
                sampler=str(rs.choice(['demc', 'snooker', 'mrw'])))
    what = int(rs.randint(0, 4))
    pstep = prob['pstep'].copy()
    prior = np.zeros(nmodel)
    plow = np.zeros(nmodel)
    pup = np.zeros(nmodel)
    if what == 0 and nmodel > 2:
        pstep[int(rs.randint(1, nmodel))] = 0.0
    elif what == 1:
        k = int(rs.randint(0, nmodel))
        prior[k], plow[k], pup[k] = p_true[k], rs.uniform(0.2, 2), rs.uniform(0.2, 2)
    elif what == 2 and prob['sampler'] != 'mrw':
        prob['fepsilon'] = 0.01
    elif prob['sampler'] != 'mrw':
        prob['sampler'] = 'mrw'
    prob.update(pstep=pstep, prior=prior, priorlow=plow, priorup=pup)
    return prob


@pytest.mark.parametrize('seed', range(40))
def test_wide_offlean_random_problem(seed):
    from mc3_amd import engine
    prob = wide_offlean_problem(seed)
    gpu = engine.DeviceChains(**prob)
    plan = gpu.plan
    label = (f"{prob['sampler']} poly{len(prob['params'])} ndata={len(prob['data'])} "
             f"nchains={prob['nchains']} served={gpu.jit_served} plan={plan}")
    assert plan['persistent'] > 0 and plan['fast'] == 2 and plan['tile'] == 0, label
    orc = H.make_oracle(prob, plan=plan)
    try:
        Z0, lp0 = H.initial_population(orc, seed=seed)
    except ValueError:
        gpu.close()
        pytest.skip('no initial population within these bounds')
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(nthreads=4)
    gpu.run(max(1, gpu.nzchain//3))
    gpu.run()
    compare_final(gpu, orc, label)
    gpu.close()


def wide_models_problem(seed):
    """The models that are specialised at run time -- polynomials of 5..8 coefficients, design
    matrices of 1..10 columns -- on 1 025 ... 16 384 points: tiles in the registers of four /
    eight waves where a thread's share fits, the tiled plans otherwise; any sampler, fixed
    coefficients, priors, the support distribution; ragged last tiles."""
    rs = np.random.RandomState(670000 + seed)
    linear = bool(rs.rand() < 0.6)
    nmodel = int(rs.randint(1, 11)) if linear else int(rs.randint(5, 9))
    nd = int(rs.choice([int(rs.randint(1025, 16385)), 1025, 2048, 2049, 4096, 4097, 5000, 8192,
                        8193, 12000]))
    nchains = int(rs.randint(3, 40))
    t = np.sort(rs.uniform(-1, 1, nd))
    p_true = rs.uniform(-1, 1, nmodel)
    if linear:
        A = np.array([np.cos(0.9*k*t + 0.2*k) for k in range(nmodel)])
        data = p_true @ A + rs.normal(0, 1, nd)
        indep, model = A, 'linear'
    else:
        data = p_true @ np.array([t**k for k in range(nmodel)]) + rs.normal(0, 1, nd)
        indep, model = t, 'poly'
    pstep = np.full(nmodel, 0.03)
    if nmodel > 2 and rs.rand() < 0.3:
        pstep[int(rs.randint(1, nmodel))] = 0.0
    prior = np.zeros(nmodel)
    plow = np.zeros(nmodel)
    pup = np.zeros(nmodel)
    if rs.rand() < 0.4:
        k = int(rs.randint(0, nmodel))
        if pstep[k] > 0:
            prior[k], plow[k], pup[k] = p_true[k], rs.uniform(0.2, 2), rs.uniform(0.2, 2)
    sampler = str(rs.choice(['demc', 'snooker', 'mrw']))
    return dict(data=data, uncert=rs.uniform(0.5, 2.0, nd), indep=indep, params=p_true.copy(),
                pstep=pstep, pmin=p_true - rs.uniform(0.5, 4.0, nmodel),
                pmax=p_true + rs.uniform(0.5, 4.0, nmodel), prior=prior, priorlow=plow,
                priorup=pup, nchains=nchains, nsamples=nchains*int(rs.randint(10, 50)),
                sampler=sampler, wlike=False, thinning=1, fgamma=float(rs.uniform(0.5, 1.5)),
                fepsilon=float(rs.choice([0.0, 0.0, 0.01])) if sampler != 'mrw' else 0.0,
                hsize=int(rs.randint(1, 4)), model=model, seed=int(rs.randint(0, 2**31)))


@pytest.mark.parametrize('seed', range(60))
def test_wide_models_random_problem(seed):
    from mc3_amd import engine
    prob = wide_models_problem(seed)
    gpu = engine.DeviceChains(**prob)
    plan = gpu.plan
    label = (f"{prob['sampler']} {prob['model']}{len(prob['params'])} ndata={len(prob['data'])} "
             f"nchains={prob['nchains']} served={gpu.jit_served} plan={plan}")
    orc = H.make_oracle(prob, plan=plan)
    try:
        Z0, lp0 = H.initial_population(orc, seed=seed)
    except ValueError:
        gpu.close()
        pytest.skip('no initial population within these bounds')
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(nthreads=4)
    gpu.run(max(1, gpu.nzchain//3))
    gpu.run()
    compare_final(gpu, orc, label)
    gpu.close()
