#!/usr/bin/env python3
"""Development aid: 256 chains x {1 000 ... 16 384 points} x polynomials of 5 / 8 coefficients
and design matrices of 5 / 10 columns x three samplers: what the models that are specialised at
run time cost beyond one register tile.   python tools/sweep_offlean_models.py > out.jsonl"""
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
    g.run(n//5)
    g.run(n)
    us = g.last_run_ms()*1e3/(n*g.thinning)
    pl = g.plan
    served = g.jit_served
    g.close()
    return us, pl, served


def polyk(nd, k, sampler, n):
    prob = H.quad_problem(ndata=nd, nchains=256, nsamples=256*n, sampler=sampler)
    pad = k - 3
    prob.update(indep=prob['indep']/10.0,
                params=np.concatenate([[10.0, -20.0, 10.0], np.zeros(pad)]),
                pstep=np.concatenate([[1.0, 5.0, 10.0], np.full(pad, 1.0)]),
                pmin=np.concatenate([[-10.0, -200.0, -1000.0], np.full(pad, -1e4)]),
                pmax=np.concatenate([[40.0, 200.0, 1000.0], np.full(pad, 1e4)]))
    return prob


for smp in ('demc', 'snooker', 'mrw'):
    for nd in (1000, 2048, 4096, 8192, 16384):
        for name, k in (('poly', 5), ('poly', 8), ('lin', 5), ('lin', 10)):
            n = 2000
            m = n + n//5 + 8
            prob = polyk(nd, k, smp, m) if name == 'poly' else \
                H.linear_problem(ndata=nd, nmodel=k, nchains=256, nsamples=256*m, sampler=smp)
            try:
                us, pl, served = measure(prob, n)
                print(json.dumps(dict(sampler=smp, ndata=nd, model=f'{name}{k}', us=round(us, 3),
                                      Msteps=round(256/us, 1), fast=pl['fast'], jit=pl['jit'],
                                      wpc=pl['waves_per_chain'], tile=pl['tile'],
                                      resident=pl['resident'], owner_waves=pl['owner_waves'])),
                      flush=True)
            except Exception as e:      # noqa: BLE001
                print(json.dumps(dict(sampler=smp, ndata=nd, model=f'{name}{k}',
                                      error=str(e)[:100])), flush=True)
