#!/usr/bin/env python3
"""Development aid: shapes around the edges of the hand-laid-out loops -- thinning beyond one
register tile, large ensembles beyond one tile, the wavelet likelihood with DEMC / mrw."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import helpers as H  # noqa: E402
import tools_pop  # noqa: E402
from mc3_amd import engine  # noqa: E402


def measure(prob, n):
    g = engine.DeviceChains(**prob)
    Z0, lp0 = tools_pop.population(g)
    g.set_population(Z0, lp0)
    g.run(max(1, n//5))
    g.run(n)
    us = g.last_run_ms()*1e3/(n*g.thinning)
    pl = g.plan
    g.close()
    return us, pl


def row(tag, prob, n):
    try:
        us, pl = measure(prob, n)
        print(json.dumps(dict(what=tag, us_per_iteration=round(us, 3),
                              Msteps=round(prob['nchains']/us, 1),
                              plan={k: pl[k] for k in ('fast', 'jit', 'waves_per_chain', 'tile',
                                                       'resident', 'chains_per_workgroup')})),
              flush=True)
    except Exception as e:      # noqa: BLE001
        print(json.dumps(dict(what=tag, error=str(e)[:120])), flush=True)


PRIOR = dict(prior=np.array([3.0, 0.0, 0.5]), priorlow=np.array([0.5, 0.0, 0.02]),
             priorup=np.array([0.25, 0.0, 0.04]))
for smp in ('demc', 'snooker', 'mrw'):
    for nd in (1000, 2048, 4096, 8192):
        n = 300
        row(f'{smp} thinning 10, {nd} points',
            H.quad_problem(ndata=nd, nchains=256, nsamples=256*(n + n//5 + 8)*10, sampler=smp,
                           thinning=10), n)
        row(f'{smp} thinning 10 + prior, {nd} points',
            H.quad_problem(ndata=nd, nchains=256, nsamples=256*(n + n//5 + 8)*10, sampler=smp,
                           thinning=10, **PRIOR), n)
for smp in ('demc', 'snooker', 'mrw'):
    for nch in (1024, 4096):
        for nd in (1000, 2048, 4096):
            n = 600
            row(f'{smp} {nch} chains, {nd} points',
                H.quad_problem(ndata=nd, nchains=nch, nsamples=nch*(n + n//5 + 8), sampler=smp), n)
            row(f'{smp} {nch} chains + prior, {nd} points',
                H.quad_problem(ndata=nd, nchains=nch, nsamples=nch*(n + n//5 + 8), sampler=smp,
                               **PRIOR), n)
for smp in ('demc', 'mrw', 'snooker'):
    for nd in (256, 2048, 8192, 16384):
        n = 1500
        row(f'wavelet {smp}, {nd} points',
            H.wavelet_problem(ndata=nd, nchains=256, nsamples=256*(n + n//5 + 8), sampler=smp), n)
