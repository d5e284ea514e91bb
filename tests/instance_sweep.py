"""The grid behind tests/test_gpu_instances.py: every combination of the things a kernel
instance of libmc3hip.so is a template of -- polynomial degree (2..4 coefficients, the generic
polynomial, a design matrix), sampler, what the parameters carry (all free / a prior and a
fixed coefficient / the support distribution), thinning, tiles, chains per wave, the declared
cadence, ranks -- as a small problem each.  The fuzz tests draw such problems at random (and so
reach the common instances many times, the rare ones never: the launch ledger showed 387 of 744
unlaunched in round 5's suite); this is the same space walked deliberately."""
import itertools

import numpy as np

import helpers as H

DEGREES = (2, 3, 4)
SAMPLERS = ('demc', 'snooker', 'mrw')
# what the parameters carry / how the run is laid out
FLAVOURS = ('plain', 'fx', 'thin', 'thinfx', 'multi', 'eps', 'epsfx')


def poly_case(ncoef, sampler, flavour, nchains=12, nref=24, ndata=None, zlag=0, seed=77):
    """The tutorial's data fitted with `ncoef` polynomial coefficients (5+: the generic
    polynomial of the general chain)."""
    thinning = 2 if flavour.startswith('thin') else 1
    if ndata is None:
        ndata = 1500 if flavour == 'multi' else 300
    prob = H.quad_problem(ndata=ndata, nchains=nchains, nsamples=nchains*nref*thinning,
                          sampler=sampler, thinning=thinning, seed=seed)
    base = np.array([10.0, -2.0, 0.1, 0.001, 0.0, 0.0])
    step = np.array([1.0, 0.5, 0.1, 0.002, 0.0002, 0.00002])
    lo = np.array([-10.0, -20.0, -10.0, -1.0, -0.1, -0.01])
    hi = np.array([40.0, 20.0, 10.0, 1.0, 0.1, 0.01])
    prob.update(params=base[:ncoef].copy(), pstep=step[:ncoef].copy(), pmin=lo[:ncoef].copy(),
                pmax=hi[:ncoef].copy(), prior=np.zeros(ncoef), priorlow=np.zeros(ncoef),
                priorup=np.zeros(ncoef))
    if sampler == 'mrw':
        prob['pstep'] = prob['pstep']*0.05
    if flavour.endswith('fx'):
        prob['prior'][0], prob['priorlow'][0], prob['priorup'][0] = 3.0, 0.5, 0.25
        if ncoef >= 3:
            prob['pstep'][ncoef - 1] = 0.0          # the last coefficient fixed
    if flavour.startswith('eps'):
        prob['fepsilon'] = 0.01
    if zlag:
        prob['zlag'] = zlag
    return prob


def linear_case(ncol, sampler, flavour, nchains=12, nref=24, ndata=300):
    thinning = 2 if flavour.startswith('thin') else 1
    prob = H.linear_problem(ndata=ndata, nmodel=ncol, nchains=nchains,
                            nsamples=nchains*nref*thinning, sampler=sampler, thinning=thinning)
    if not flavour.endswith('fx'):
        for key in ('prior', 'priorlow', 'priorup'):
            prob[key] = np.zeros(ncol)
    if flavour.startswith('eps'):
        prob['fepsilon'] = 0.01
    return prob


def wavelet_case(ncoef, sampler, ndata, nchains=8, nref=16, flavour='plain'):
    """Wavelet likelihood behind a polynomial of `ncoef` coefficients (the last three parameters
    are the noise model's: gamma fixed, sigma_r and sigma_w free)."""
    thinning = 2 if flavour.startswith('thin') else 1
    prob = H.wavelet_problem(ndata=ndata, nchains=nchains, nsamples=nchains*nref*thinning,
                             sampler=sampler, thinning=thinning)
    base = np.array([3.0, -2.4, 0.5, 0.0, 0.0, 0.0])
    step = np.array([0.3, 0.1, 0.02, 0.002, 0.0002, 0.00002])
    lo = np.array([-10.0, -20.0, -10.0, -1.0, -0.1, -0.01])
    hi = np.array([40.0, 20.0, 10.0, 1.0, 0.1, 0.01])
    noise = dict(params=[1.0, 0.5, 1.0], pstep=[0.0, 0.1, 0.1], pmin=[0.0, 0.01, 0.01],
                 pmax=[2.0, 10.0, 10.0])
    for key, src in (('params', base), ('pstep', step), ('pmin', lo), ('pmax', hi)):
        prob[key] = np.concatenate([src[:ncoef], noise[key]])
    if sampler == 'mrw':
        prob['pstep'] = prob['pstep']*0.2
    if flavour.endswith('fx'):
        n = ncoef + 3
        prob.update(prior=np.zeros(n), priorlow=np.zeros(n), priorup=np.zeros(n))
        prob['prior'][0], prob['priorlow'][0], prob['priorup'][0] = 3.0, 0.5, 0.25
    return prob


def wavelet_linear_case(sampler, ndata, nchains=8, nref=16, flavour='plain'):
    """the same with the model as a design matrix of three columns (1, x, x^2)"""
    prob = wavelet_case(3, sampler, ndata, nchains=nchains, nref=nref, flavour=flavour)
    x = prob['indep']
    prob.update(model='linear', nmodel=3, indep=np.array([np.ones_like(x), x, x*x]))
    return prob


def one_gpu_lean_grid():
    """(label, problem, chains per wave) over degree x sampler x flavour x chains per wave."""
    # (five chains per wave: beyond what the rows loops lay side by side -- the lean instances
    # of run_chain take turns)
    for ncoef, sampler, flavour, cpw in itertools.product(DEGREES, SAMPLERS, FLAVOURS,
                                                          (1, 2, 3, 4, 5)):
        if flavour.startswith('eps') and sampler == 'mrw':
            continue                                     # (chain.py:219-220: no such thing)
        yield (f'poly{ncoef}-{sampler}-{flavour}-cpw{cpw}',
               poly_case(ncoef, sampler, flavour, nchains=12 if cpw == 1 else 4*cpw + 1), cpw)


def one_gpu_big_ensembles_grid():
    """ensembles beyond one wave per SIMD (the draws of 64 refreshes at once), and more than one
    tile with too many chains for the tiles to sit in the registers of four waves each"""
    for ncoef in DEGREES:
        # (more than 1 024 chains: lean_demc_kernel BIG, the wait laid out by hand; the run crosses
        # a batch of 64 refreshes' draws)
        yield (f'poly{ncoef}-demc-batch', poly_case(ncoef, 'demc', 'plain', nchains=1100, nref=70,
                                                    ndata=100), 1)
        for sampler in SAMPLERS:
            yield (f'poly{ncoef}-{sampler}-multi-800',
                   poly_case(ncoef, sampler, 'multi', nchains=800, nref=4), 1)


def one_gpu_lag_grid():
    """the declared cadence on one GPU: zlag 1 and 3, with and without thinning"""
    for ncoef, flavour, cpw, zlag in itertools.product(DEGREES, ('plain', 'fx', 'thin', 'thinfx'),
                                                       (1, 2, 3, 4), (1, 3)):
        yield (f'poly{ncoef}-demc-{flavour}-cpw{cpw}-zlag{zlag}',
               poly_case(ncoef, 'demc', flavour, nchains=12 if cpw == 1 else 4*cpw + 1,
                         zlag=zlag), cpw)


# ---- sharded over ranks (all on one device in the tests): the peer-to-peer flavours ----------
def ranks_lean_grid(nranks):
    """(label, problem, options) -- degree x sampler x flavour x (courier wave | the chain wave
    stores its rows itself) x chains per wave; strict lockstep and the declared cadence."""
    for ncoef, sampler, flavour in itertools.product(DEGREES, SAMPLERS,
                                                     ('plain', 'fx', 'thin', 'thinfx')):
        for no_courier in (False, True):
            yield (f'poly{ncoef}-{sampler}-{flavour}-{"nocourier" if no_courier else "courier"}',
                   poly_case(ncoef, sampler, flavour, nchains=6*nranks, nref=18),
                   dict(no_courier=no_courier))
        if flavour in ('plain', 'fx'):
            for cpw in (2, 3, 4):                       # the chains of a wave side by side
                per = 4*cpw + 1
                yield (f'poly{ncoef}-{sampler}-{flavour}-cpw{cpw}',
                       poly_case(ncoef, sampler, flavour, nchains=per*nranks, nref=18),
                       dict(no_courier=True, env=dict(MC3HIP_PERSIST_CAPACITY=-(-per//cpw))))
    for ncoef, flavour, zlag in itertools.product(DEGREES, ('plain', 'fx'), (2,)):
        for no_courier in (False, True):
            yield (f'poly{ncoef}-demc-{flavour}-zlag{zlag}-'
                   f'{"nocourier" if no_courier else "courier"}',
                   poly_case(ncoef, 'demc', flavour, nchains=6*nranks, nref=18, zlag=zlag),
                   dict(no_courier=no_courier))
        for cpw in (2, 3, 4):
            per = 4*cpw + 1
            yield (f'poly{ncoef}-demc-{flavour}-zlag{zlag}-cpw{cpw}',
                   poly_case(ncoef, 'demc', flavour, nchains=per*nranks, nref=18, zlag=zlag),
                   dict(no_courier=True, env=dict(MC3HIP_PERSIST_CAPACITY=-(-per//cpw))))


def ranks_general_grid(nranks):
    """what no hand-laid-out loop serves, sharded: the general chain's peer-to-peer instances --
    generic polynomials and design matrices (the run-time specialisation is a one-GPU affair),
    the support distribution, the wavelet likelihood on one and four waves per chain and beyond
    2^14 points -- by tiles and waves per chain"""
    for sampler, ndata in itertools.product(SAMPLERS, (300, 1500, 5000)):
        yield (f'poly5-{sampler}-{ndata}', poly_case(5, sampler, 'fx', nchains=4*nranks, nref=12,
                                                     ndata=ndata), {})
        yield (f'lin7-{sampler}-{ndata}', linear_case(7, sampler, 'fx', nchains=4*nranks, nref=12,
                                                     ndata=ndata), {})
    for ncoef, sampler, ndata in itertools.product(DEGREES, ('demc', 'snooker'), (300, 1500, 5000)):
        yield (f'poly{ncoef}-{sampler}-eps-{ndata}',
               poly_case(ncoef, sampler, 'eps', nchains=4*nranks, nref=12, ndata=ndata), {})
    for k, sampler, ndata in itertools.product((2, 3, 4, 5, 0), SAMPLERS, (256, 4096, 32768)):
        if ndata == 32768 and sampler != 'snooker':
            continue
        prob = (wavelet_case(k, sampler, ndata, nchains=2*nranks, nref=8) if k else
                wavelet_linear_case(sampler, ndata, nchains=2*nranks, nref=8))
        yield (f'w-{"poly%d" % k if k else "lin3"}-{sampler}-{ndata}', prob, {})


def ranks_resident_grid(nranks):
    """the one-launch tiled plans sharded over ranks: design matrices by blocks of operand rows
    (3, 7, 10, 13 columns), polynomials of 2..4 coefficients; the hand-structured owner (demc) and
    the general chain as owner (snooker); the kernel in step and -- columns of at least six
    (four) blocks of sixteen chains, three chains to own per workgroup -- with nobody in step"""
    for ncol, sampler in itertools.product((3, 7, 10, 13), ('demc', 'snooker')):
        yield (f'res-lin{ncol}-{sampler}',
               linear_case(ncol, sampler, 'fx', nchains=32*nranks, nref=8, ndata=20000),
               dict(env=dict(MC3HIP_RESIDENT_MAXCOL=2)))
    for ncoef, sampler in itertools.product(DEGREES, ('demc', 'snooker')):
        yield (f'res-poly{ncoef}-{sampler}',
               poly_case(ncoef, sampler, 'fx', nchains=32*nranks, nref=8, ndata=20000),
               dict(env=dict(MC3HIP_RESIDENT_MAXCOL=2)))
    for ncol, sampler in ((3, 'demc'), (7, 'demc'), (10, 'demc'), (13, 'demc'), (3, 'snooker'),
                          (7, 'snooker'), (10, 'snooker')):
        yield (f'res-async-lin{ncol}-{sampler}',
               linear_case(ncol, sampler, 'fx', nchains=96*nranks, nref=6, ndata=70000),
               dict(env=dict(MC3HIP_RESIDENT_MAXCOL=1)))


GRIDS = {'lean': ranks_lean_grid, 'general': ranks_general_grid, 'resident': ranks_resident_grid}


# ---- the general chain (run_chain behind persist_kernel / step_kernel) ------------------------
def general_grid():
    """(label, problem, options): what no hand-laid-out loop serves -- the per-iteration trace
    (polynomials of 2..4 coefficients), the generic polynomial and design matrices with the
    run-time specialisation switched off -- by waves per chain (1: up to 4 096 points, 4:
    beyond), tiles, chains per workgroup, launch structure (persistent / one launch per
    refresh)."""
    models = [(f'poly{k}', lambda smp, nd, nc, k=k: poly_case(k, smp, 'plain', nchains=nc, nref=10,
                                                            ndata=nd), dict(trace=True))
              for k in DEGREES]
    models.append(('poly5', lambda smp, nd, nc: poly_case(5, smp, 'fx', nchains=nc, nref=10,
                                                        ndata=nd), dict(jit=False)))
    models.append(('lin7', lambda smp, nd, nc: linear_case(7, smp, 'fx', nchains=nc, nref=10,
                                                         ndata=nd), dict(jit=False)))
    for (name, build, opt), sampler, ndata in itertools.product(models, SAMPLERS,
                                                                (300, 1500, 5000)):
        for cpw, no_persist in ((1, False), (2, False), (1, True)):
            yield (f'{name}-{sampler}-{ndata}-cpw{cpw}{"-perrefresh" if no_persist else ""}',
                   build(sampler, ndata, 9 if cpw == 2 else 8),
                   dict(opt, cpw=cpw, no_persist=no_persist))


def wavelet_grid():
    """wavelet likelihood: the model behind it x sampler x geometry (one wave per chain up to
    2 048 points, four at 4 096, eight and a chain wave above, the large-data instances beyond
    2^14) x chains per workgroup x launch structure; `thin`: the general chain as chain wave"""
    models = [(f'poly{k}', k) for k in (2, 3, 4, 5)] + [('lin3', 0)]
    for (name, k), sampler, ndata in itertools.product(models, SAMPLERS,
                                                       (256, 4096, 8192, 16384, 32768)):
        for flavour, cpw, no_persist in (('plain', 1, False), ('plain', 2, False),
                                         ('plain', 1, True), ('thin', 1, False),
                                         ('fx', 1, False)):
            if ndata >= 16384 and (flavour != 'plain' or (sampler == 'mrw' and cpw == 1
                                                          and no_persist)):
                continue
            yield (f'w-{name}-{sampler}-{ndata}-{flavour}-cpw{cpw}'
                   f'{"-perrefresh" if no_persist else ""}',
                   (wavelet_case(k, sampler, ndata, nchains=5 if cpw == 2 else 4,
                                 nref=6 if ndata >= 8192 else 10, flavour=flavour) if k else
                    wavelet_linear_case(sampler, ndata, nchains=5 if cpw == 2 else 4,
                                        nref=6 if ndata >= 8192 else 10, flavour=flavour)),
                   dict(cpw=cpw, no_persist=no_persist))


def tiled_grid():
    """the three-launch tiled plan (proposal / tile sums / finish) on the vector unit: polynomials
    of 4..16 coefficients and one- and sixteen-column design matrices at 20 000 points, the
    run-time specialisation and the one-launch plan switched off"""
    for name, prob in (
            [(f'poly{k}', poly_case(k, smp, 'plain', nchains=16, nref=6, ndata=20000))
             for k, smp in ((4, 'demc'), (4, 'snooker'))] +
            [(f'lin{k}', linear_case(k, smp, 'fx', nchains=16, nref=6, ndata=20000))
             for k, smp in ((1, 'demc'), (6, 'snooker'), (10, 'mrw'), (13, 'demc'),
                            (16, 'demc'), (16, 'snooker'))]):
        for no_persist in (False, True):
            yield (f'tiled-{name}-{prob["sampler"]}{"-threelaunch" if no_persist else ""}', prob,
                   dict(jit=False, no_persist=no_persist))
    # generic polynomials of 6, 10, 14 coefficients: the vector-unit tile kernels
    for k in (6, 10, 14):
        prob = poly_case(5, 'demc', 'plain', nchains=16, nref=6, ndata=20000)
        x = prob['indep']/10.0
        prob.update(indep=x, params=np.concatenate([prob['params'], np.zeros(k - 5)]),
                    pstep=np.concatenate([prob['pstep'], np.full(k - 5, 1e-4)]),
                    pmin=np.concatenate([prob['pmin'], np.full(k - 5, -1.0)]),
                    pmax=np.concatenate([prob['pmax'], np.full(k - 5, 1.0)]),
                    prior=np.zeros(k), priorlow=np.zeros(k), priorup=np.zeros(k))
        yield (f'tiled-poly{k}-demc', prob, dict(jit=False))
