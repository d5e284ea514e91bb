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
    'c3onesnooker': lambda n: (H.quad_problem(ndata=1000, nchains=2048, nsamples=2048*n,
                                              sampler='snooker'), {}),
    'c4x2': lambda n: (H.wavelet_problem(ndata=16384, nchains=512, nsamples=512*n), {}),
    'c4x4': lambda n: (H.wavelet_problem(ndata=16384, nchains=1024, nsamples=1024*n), {}),
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
CONFIGS['p6'] = lambda n: _poly6(n, False)
CONFIGS['p6matrix'] = lambda n: _poly6(n, True)
BYTES = {'c2mrw': 24104, 'q5': 8*3*100000 + 8*3*4 + 8, 'c4mrw': 8*2*16384 + 8*5*4 + 8, 'c4demc': 8*2*16384 + 8*5*4 + 8, 'c4half': 8*2*16384 + 8*5*4 + 8, 'c2': 24104, 'c2snooker': 24104, 'c3one': 24104, 'c4': 8*2*16384 + 8*5*4 + 8,
         'c5': 8*12*100000 + 8*10*4 + 8, 'c5share': 8*12*100000 + 8*10*4 + 8}
ITERS = {'c2': 20000, 'c2snooker': 20000, 'c2mrw': 20000, 'c3one': 4000, 'c4': 4000, 'c5': 30, 'c5share': 1}
ITERS['c5'] = 300
ITERS['q5'] = 300
ITERS.update(c4big=1000, c4wide=4000, c4mrw=4000, c4demc=4000, c4half=4000, c3x2=3000, c3x4=2000, c3x2snooker=3000,
             c4x2=2000, c4x4=1000)
ITERS.update(c3x2mrw=3000, c3onesnooker=4000, p6=200, p6matrix=200, c3x4snooker=1500)
BYTES['c3x4snooker'] = 24104
for _smp in ('demc', 'snooker', 'mrw'):
    ITERS['c2thin10' + _smp] = 4000
    BYTES['c2thin10' + _smp] = 24104
    ITERS['c2two' + _smp] = 10000
    BYTES['c2two' + _smp] = 48104
BYTES['p6'] = BYTES['p6matrix'] = 8*3*100000 + 8*6*4 + 8
for _k in ('c3x2', 'c3x4', 'c3x2snooker', 'c3x2mrw', 'c3onesnooker'):
    BYTES[_k] = 24104
BYTES['c4big'] = 8*2*65536 + 8*5*4 + 8
for _k in ('c4x2', 'c4x4', 'c4wide'):
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
            'plan': g.plan, 'acceptance': round(float(acc), 3),
            'out_of_bounds_max_param': round(float(oob), 3)}), flush=True)
        g.close()


if __name__ == '__main__':
    main()
