#!/usr/bin/env python3
"""Throughput of the BASELINE.json configurations other than the headline one
(bench.py measures C2).  Not the judged bench line: a development tool whose
output goes to profiles/.

    python tools/bench_configs.py [c2 c3share c4 c5 c5share ...]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mc3_amd import engine  # noqa: E402
import helpers as H  # noqa: E402


def population(g, prob, seed=5):
    """In-bounds draws around the start with finite chisq, evaluated on the device."""
    rs = np.random.RandomState(seed)
    ifree = g.ifree
    Z0 = np.empty((0, g.nfree))
    lp0 = np.empty(0)
    while len(Z0) < g.M0:
        trial = rs.normal(g.params[ifree], g.pstep[ifree], (2*g.M0, g.nfree))
        full = g.expand_free(trial)
        ok = np.all((full >= g.pmin) & (full <= g.pmax), axis=1)
        lp = -0.5*g.eval_chisq(full[ok])
        fin = np.isfinite(lp)
        Z0 = np.concatenate([Z0, trial[ok][fin]])
        lp0 = np.concatenate([lp0, lp[fin]])
    return np.ascontiguousarray(Z0[:g.M0]), np.ascontiguousarray(lp0[:g.M0])


CONFIGS = {
    'c2': lambda n: (H.quad_problem(ndata=1000, nchains=256, nsamples=256*n, sampler='demc'), {}),
    'c2snooker': lambda n: (H.quad_problem(ndata=1000, nchains=256, nsamples=256*n,
                                           sampler='snooker'), {}),
    'c2mrw': lambda n: (H.quad_problem(ndata=1000, nchains=256, nsamples=256*n, sampler='mrw'), {}),
    'c3one': lambda n: (H.quad_problem(ndata=1000, nchains=2048, nsamples=2048*n,
                                       sampler='demc'), {}),
    'c4': lambda n: (H.wavelet_problem(ndata=16384, nchains=256, nsamples=256*n), {}),
    # beyond what is co-resident at one chain per workgroup: several chains per workgroup
    'c3x2': lambda n: (H.quad_problem(ndata=1000, nchains=4096, nsamples=4096*n,
                                      sampler='demc'), {}),
    'c3x4': lambda n: (H.quad_problem(ndata=1000, nchains=8192, nsamples=8192*n,
                                      sampler='demc'), {}),
    'c3x2snooker': lambda n: (H.quad_problem(ndata=1000, nchains=4096, nsamples=4096*n,
                                             sampler='snooker'), {}),
    'c3x2mrw': lambda n: (H.quad_problem(ndata=1000, nchains=4096, nsamples=4096*n,
                                         sampler='mrw'), {}),
    'c3x4snooker': lambda n: (H.quad_problem(ndata=1000, nchains=8192, nsamples=8192*n,
                                             sampler='snooker'), {}),
    'c3x4mrw': lambda n: (H.quad_problem(ndata=1000, nchains=8192, nsamples=8192*n,
                                         sampler='mrw'), {}),
    'c3onesnooker': lambda n: (H.quad_problem(ndata=1000, nchains=2048, nsamples=2048*n,
                                              sampler='snooker'), {}),
    'c4x2': lambda n: (H.wavelet_problem(ndata=16384, nchains=512, nsamples=512*n), {}),
    'c4x4': lambda n: (H.wavelet_problem(ndata=16384, nchains=1024, nsamples=1024*n), {}),
    'c4x2wide': lambda n: (H.wavelet_problem(ndata=16384, nchains=512, nsamples=512*n,
                                             pmin=np.full(6, -1e3), pmax=np.full(6, 1e3)), {}),
    'c4x2mrw': lambda n: (H.wavelet_problem(ndata=16384, nchains=512, nsamples=512*n,
                                            sampler='mrw'), {}),
    'c4mrw': lambda n: (H.wavelet_problem(ndata=16384, nchains=256, nsamples=256*n,
                                          sampler='mrw'), {}),
    'c4demc': lambda n: (H.wavelet_problem(ndata=16384, nchains=256, nsamples=256*n,
                                           sampler='demc'), {}),
    # no bounds to speak of: every proposal is evaluated (C4's own bounds reject 3 in 10
    # snooker proposals before the likelihood, as the reference does)
    'c4wide': lambda n: (H.wavelet_problem(ndata=16384, nchains=256, nsamples=256*n,
                                           pmin=np.full(6, -1e3), pmax=np.full(6, 1e3)), {}),
    # beyond 2^14 points: pre-levels through global memory, then the blocked pyramid
    'c4big': lambda n: (H.wavelet_problem(ndata=65536, nchains=256, nsamples=256*n), {}),
    'c4half': lambda n: (H.wavelet_problem(ndata=16384, nchains=128, nsamples=128*n), {}),
    'c5': lambda n: (H.linear_problem(ndata=100000, nchains=512, nsamples=512*n), {}),
    'c5snooker': lambda n: (H.linear_problem(ndata=100000, nchains=512, nsamples=512*n,
                                             sampler='snooker'), {}),
    'c5mrw': lambda n: (H.linear_problem(ndata=100000, nchains=512, nsamples=512*n,
                                         sampler='mrw'), {}),
    'c5share': lambda n: (H.linear_problem(ndata=100000, nchains=512, nsamples=512*n),
                          dict(chain0=0, nlocal=128)),
    # the tiled plan of polynomial models (vector tile kernel, three launches per iteration)
    'q5': lambda n: (H.quad_problem(ndata=100000, nchains=512, nsamples=512*n), {}),
}


def _poly6(n, as_matrix):
    """A polynomial of six coefficients x 1e5 points x 512 chains, as the device's polynomial
    family (three-launch tiled plan) or as the design matrix of the powers of x that
    mc3_amd.models.lower() hands over (one-launch plan of the matrix cores)."""
    prob = H.linear_problem(ndata=100000, nmodel=6, nchains=512, nsamples=512*n)
    x = np.linspace(-1.0, 1.0, 100000)
    if as_matrix:
        prob.update(indep=np.array([x**k for k in range(6)]))
    else:
        prob.update(indep=x, model='poly')
    return prob, {}


# thinning > 1: the iterations between two refreshes look at the same rows
for _smp in ('demc', 'snooker', 'mrw'):
    CONFIGS['c2thin10' + _smp] = (lambda n, _s=_smp: (H.quad_problem(
        ndata=1000, nchains=256, nsamples=256*n*10, sampler=_s, thinning=10), {}))
# two tiles (streamed-tile instances)
for _smp in ('demc', 'snooker', 'mrw'):
    CONFIGS['c2two' + _smp] = (lambda n, _s=_smp: (H.quad_problem(
        ndata=2000, nchains=256, nsamples=256*n, sampler=_s), {}))
# ---- round 4: the shapes that fall off the hand-laid-out loops (VERDICT r3 row g), all at the
# headline size -- 256 chains x 1000 points (mc3/chain.py:217,232 epsilon; :246-247 shared
# parameters; stats.py:211 priors)
def _off(n, sampler='demc', thinning=1, **kw):
    return H.quad_problem(ndata=1000, nchains=256, nsamples=256*n*thinning, sampler=sampler,
                          thinning=thinning, **kw), {}


def _polyk(n, k, sampler='demc'):
    # a polynomial of k coefficients around the tutorial's quadratic (higher orders ~ 0)
    prob, _ = _off(n, sampler)
    x = prob['indep']/10.0
    pad = k - 3
    prob.update(indep=x,
                params=np.concatenate([[10.0, -20.0, 10.0], np.zeros(pad)]),
                pstep=np.concatenate([[1.0, 5.0, 10.0], np.full(pad, 1.0)]),
                pmin=np.concatenate([[-10.0, -200.0, -1000.0], np.full(pad, -1e4)]),
                pmax=np.concatenate([[40.0, 200.0, 1000.0], np.full(pad, 1e4)]))
    return prob, {}


def _link(n, k, sampler='demc'):
    return H.linear_problem(ndata=1000, nmodel=k, nchains=256, nsamples=256*n,
                            sampler=sampler), {}


def _with(pair, **kw):
    pair[0].update(kw)
    return pair


def _shared(n, sampler='demc', thinning=1):
    # the quadratic plus a cubic coefficient shared with the linear one (pstep = -2) -- and
    # tiny in effect: x is scaled to [0, 1]
    prob, _ = _off(n, sampler, thinning)
    prob.update(indep=prob['indep']/10.0, params=np.array([10.0, -20.0, 10.0, -20.0]),
                pstep=np.array([1.0, 5.0, 10.0, -2.0]),
                pmin=np.array([-10.0, -200.0, -1000.0, -200.0]),
                pmax=np.array([40.0, 200.0, 1000.0, 200.0]))
    return prob, {}


_PRIOR = dict(prior=np.array([3.0, 0.0, 0.5]), priorlow=np.array([0.5, 0.0, 0.02]),
              priorup=np.array([0.25, 0.0, 0.04]))
OFF_LEAN = {
    'off_shared': lambda n: _shared(n),
    'off_eps': lambda n: _off(n, fepsilon=0.01),
    'off_eps_snooker': lambda n: _off(n, 'snooker', fepsilon=0.01),
    'off_poly5': lambda n: _polyk(n, 5),
    'off_poly8': lambda n: _polyk(n, 8),
    'off_lin5': lambda n: _link(n, 5),
    'off_lin10': lambda n: _link(n, 10),
    'off_prior_thin10': lambda n: _off(n, thinning=10, **_PRIOR),
    'off_prior_snooker': lambda n: _off(n, 'snooker', **_PRIOR),
    'off_prior': lambda n: _off(n, **_PRIOR),
    'off_shared_thin10': lambda n: _shared(n, thinning=10),
    'off_poly5_snooker': lambda n: _polyk(n, 5, 'snooker'),
    'off_lin10_snooker': lambda n: _link(n, 10, 'snooker'),
    # round 5: the random walk and the support distribution's deviates beyond four coefficients
    'off_poly5_mrw': lambda n: _polyk(n, 5, 'mrw'),
    'off_poly8_mrw': lambda n: _polyk(n, 8, 'mrw'),
    'off_lin10_mrw': lambda n: _link(n, 10, 'mrw'),
    'off_poly6_eps': lambda n: _with(_polyk(n, 6), fepsilon=0.01),
    'off_lin10_eps': lambda n: _with(_link(n, 10), fepsilon=0.01),
    'off_poly6_eps_snooker': lambda n: _with(_polyk(n, 6, 'snooker'), fepsilon=0.01),
    'off_lin10_eps_snooker': lambda n: _with(_link(n, 10, 'snooker'), fepsilon=0.01),
}
CONFIGS.update(OFF_LEAN)
# the declared multi-GPU cadence on one GPU (zlag 4)
CONFIGS['c2lag4'] = lambda n: _off(n, zlag=4)
CONFIGS['c2lag1'] = lambda n: _off(n, zlag=1)
CONFIGS['c2lag4snooker'] = lambda n: _off(n, 'snooker', zlag=4)
CONFIGS['c3onelag4'] = lambda n: (H.quad_problem(ndata=1000, nchains=2048, nsamples=2048*n,
                                                 sampler='demc', zlag=4), {})
CONFIGS['p6'] = lambda n: _poly6(n, False)
CONFIGS['p6matrix'] = lambda n: _poly6(n, True)
BYTES = {'c2mrw': 24104, 'q5': 8*3*100000 + 8*3*4 + 8, 'c4mrw': 8*2*16384 + 8*5*4 + 8, 'c4demc': 8*2*16384 + 8*5*4 + 8, 'c4half': 8*2*16384 + 8*5*4 + 8, 'c2': 24104, 'c2snooker': 24104, 'c3one': 24104, 'c4': 8*2*16384 + 8*5*4 + 8,
         'c5': 8*12*100000 + 8*10*4 + 8, 'c5snooker': 8*12*100000 + 8*10*4 + 8,
         'c5mrw': 8*12*100000 + 8*10*4 + 8, 'c5share': 8*12*100000 + 8*10*4 + 8}
ITERS = {'c2': 20000, 'c2snooker': 20000, 'c2mrw': 20000, 'c3one': 4000, 'c4': 4000, 'c5': 30, 'c5share': 1}
ITERS['c5'] = ITERS['c5snooker'] = ITERS['c5mrw'] = 300
ITERS['q5'] = 300
ITERS.update(c4big=1000, c4wide=4000, c4mrw=4000, c4demc=4000, c4half=4000, c3x2=3000, c3x4=2000, c3x2snooker=3000,
             c4x2=2000, c4x4=1000, c4x2wide=2000, c4x2mrw=2000)
ITERS.update(c3x2mrw=3000, c3onesnooker=4000, p6=200, p6matrix=200, c3x4snooker=1500,
             c3x4mrw=1500)
BYTES['c3x4snooker'] = BYTES['c3x4mrw'] = 24104
for _smp in ('demc', 'snooker', 'mrw'):
    ITERS['c2thin10' + _smp] = 4000
    BYTES['c2thin10' + _smp] = 24104
    ITERS['c2two' + _smp] = 10000
    BYTES['c2two' + _smp] = 48104
BYTES['p6'] = BYTES['p6matrix'] = 8*3*100000 + 8*6*4 + 8
for _k in ('c3x2', 'c3x4', 'c3x2snooker', 'c3x2mrw', 'c3onesnooker'):
    BYTES[_k] = 24104
for _k in OFF_LEAN:
    ITERS[_k] = 3000 if 'thin10' in _k else 12000
    BYTES[_k] = 24104
BYTES.update(off_poly5=8*3*1000 + 8*5*4 + 8, off_poly8=8*3*1000 + 8*8*4 + 8,
             off_lin5=8*7*1000 + 8*5*4 + 8, off_lin10=8*12*1000 + 8*10*4 + 8,
             off_poly5_snooker=8*3*1000 + 8*5*4 + 8, off_lin10_snooker=8*12*1000 + 8*10*4 + 8)
for _k in ('c2lag4', 'c2lag1', 'c2lag4snooker'):
    ITERS[_k] = 20000
    BYTES[_k] = 24104
ITERS['c3onelag4'] = 4000
BYTES['c3onelag4'] = 24104
CONFIGS['c3x4lag4'] = lambda n: (H.quad_problem(ndata=1000, nchains=8192, nsamples=8192*n,
                                                sampler='demc', zlag=4), {})
CONFIGS['c3x2lag4'] = lambda n: (H.quad_problem(ndata=1000, nchains=4096, nsamples=4096*n,
                                                sampler='demc', zlag=4), {})
ITERS.update(c3x4lag4=2000, c3x2lag4=3000)
BYTES['c3x4lag4'] = BYTES['c3x2lag4'] = 24104
BYTES['c4big'] = 8*2*65536 + 8*5*4 + 8
for _k in ('c4x2', 'c4x4', 'c4wide', 'c4x2wide', 'c4x2mrw'):
    BYTES[_k] = BYTES['c4']


def main():
    names = sys.argv[1:] or ['c2', 'c2snooker', 'c3one', 'c4', 'c5']
    if len(names) > 1:
        # one fresh process per configuration: after the multi-GB allocations of an earlier
        # configuration have been freed, the streaming kernels of C5 were seen to run 2-3x
        # slower in the same process (placement of the new allocations)
        import subprocess
        for name in names:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), name])
        return
    for name in names:
        n = ITERS[name]
        prob, extra = CONFIGS[name](n + max(2, n//10))
        g = engine.DeviceChains(**prob, **extra)
        Z0, lp0 = population(g, prob)
        g.set_population(Z0, lp0)
        if name != 'c5share':
            g.run(max(2, n//10))
        t0 = time.perf_counter()
        g.run(n)
        dt = time.perf_counter() - t0
        steps = g.nlocal*n*g.thinning
        rate = steps/dt/1e6
        state = g.state()
        acc = state['numaccept'].sum()/(g.nlocal*g.refresh_done*g.thinning)
        oob = np.max(state['outbounds'])/(g.nlocal*g.refresh_done*g.thinning)
        print(json.dumps({
            'config': name, 'nchains': g.nchains, 'nlocal': g.nlocal, 'ndata': g.ndata,
            'sampler': g.sampler, 'wlike': g.wlike, 'iterations': n,
            'million_chain_steps_per_s': round(rate, 3),
            'us_per_iteration': round(dt/n*1e6, 2),
            'algorithmic_GBps': round(rate*1e6*BYTES[name]/1e9, 1),
            'frac_of_8TBps': round(rate*1e6*BYTES[name]/8e12, 4),
            # design-matrix models: the contraction is what bounds them (FP64 matrix / vector
            # pipe, 78.6 TFLOP/s dense on MI355X), not the chain-invariant operand streams
            **({'useful_fp64_TFLOPs': round(rate*1e6*2*g.nmodel*g.ndata/1e12, 2),
                'frac_of_fp64_peak': round(rate*1e6*2*g.nmodel*g.ndata/78.6e12, 4)}
               if name.startswith('c5') else {}),
            # the roofline of C4 (its byte figure is void: the chain-invariant streams sit in
            # LDS): FP64 vector issue.  2 335 FP64 mul / fma wave-instructions per chain-step at
            # 2^14 points (SQ_INSTS_VALU_FMA_F64 + _MUL_F64 of profiles/r03_c4_pmc.json / 4000
            # iterations / 256 chains; adds not counted), 4 cycles each, 1024 SIMDs at 2.4 GHz
            **({'fp64_issue_frac': round(rate*1e6*2335.0*(g.ndata/16384.0)*4/(1024*2.4e9), 4)}
               if name.startswith('c4') else {}),
            'plan': g.plan, 'acceptance': round(float(acc), 3),
            'out_of_bounds_max_param': round(float(oob), 3)}), flush=True)
        g.close()


if __name__ == '__main__':
    main()
