#!/usr/bin/env python3
"""One-off soak: a long run on the GPU against the oracle (OpenMP on the host cores), compared
exactly.  python tests/soak.py [iterations] [sampler] [nchains] [wavelet ndata] [key=value ...]
(C2's shape by default; `wavelet 16384` = C4's; round 4: zlag=4 fepsilon=0.01 thinning=3 fx=1
poly=8 lin=10 pick the declared cadence / the EPS loops / the run-time specialisations)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mc3_amd import engine  # noqa: E402
import helpers as H  # noqa: E402

# key=value extras (round 4): zlag=4 fepsilon=0.01 thinning=3 fx=1 poly=8 lin=10
extras = dict(a.split('=') for a in sys.argv[1:] if '=' in a)
argv = [a for a in sys.argv if '=' not in a]
niter = int(argv[1]) if len(argv) > 1 else 200000
sampler = argv[2] if len(argv) > 2 else 'demc'
nchains = int(argv[3]) if len(argv) > 3 else 256
kw = {k: (float(v) if k == 'fepsilon' else int(v)) for k, v in extras.items()
      if k in ('zlag', 'fepsilon', 'thinning')}
thin = kw.get('thinning', 1)
if len(argv) > 5 and argv[4] == 'wavelet':
    prob = H.wavelet_problem(ndata=int(argv[5]), nchains=nchains, nsamples=nchains*niter,
                             sampler=sampler, **kw)
elif 'lin' in extras:
    # (ndata=100000 nchains=512 lin=10 fx=1: C5, the resident tiled plan with nobody in step)
    prob = H.linear_problem(ndata=int(extras.get('ndata', 1000)), nmodel=int(extras['lin']),
                            nchains=nchains,
                            nsamples=nchains*niter*thin, sampler=sampler, **kw)
    if not int(extras.get('fx', 0)):
        for key in ('prior', 'priorlow', 'priorup'):
            prob[key] = np.zeros(int(extras['lin']))
elif 'poly' in extras:
    k = int(extras['poly'])
    prob = H.quad_problem(ndata=1000, nchains=nchains, nsamples=nchains*niter*thin,
                          sampler=sampler, **kw)
    prob.update(indep=prob['indep']/10.0,
                params=np.concatenate([[10.0, -20.0, 10.0], np.zeros(k - 3)]),
                pstep=np.concatenate([[1.0, 5.0, 10.0], np.full(k - 3, 1.0)]),
                pmin=np.concatenate([[-10.0, -200.0, -1000.0], np.full(k - 3, -1e4)]),
                pmax=np.concatenate([[40.0, 200.0, 1000.0], np.full(k - 3, 1e4)]))
elif int(extras.get('fx', 0)):
    prob = H.lean_shape_problem(nchains, niter, ndata=1000, sampler=sampler, fx=1, **kw)
else:
    # (ndata=8192: the tiles in the registers of several waves, round 5)
    prob = H.quad_problem(ndata=int(extras.get('ndata', 1000)), nchains=nchains,
                          nsamples=nchains*niter*thin, sampler=sampler, **kw)
gpu = engine.DeviceChains(**prob)
orc = H.make_oracle(prob, plan=gpu.plan)    # (the plan: the order a tiled plan adds up in)
Z0, lp0 = H.initial_population(orc)
orc.set_population(Z0, lp0)
gpu.set_population(Z0, lp0)
t0 = time.perf_counter()
gpu.run()
tg = time.perf_counter() - t0
nthreads = min(64, len(os.sched_getaffinity(0)))
t0 = time.perf_counter()
orc.run(nthreads=nthreads)
to = time.perf_counter() - t0
Z, lp, zc = gpu.rows()
st = gpu.state()
same = (np.array_equal(Z, orc.Z) and np.array_equal(lp, orc.log_post)
        and np.array_equal(zc, orc.zchain)
        and np.array_equal(st['numaccept'], orc.numaccept)
        and np.array_equal(st['outbounds'], orc.outbounds)
        and np.array_equal(st['best_chisq'], orc.best_chisq))
niter *= thin
print(f'{sampler} {extras}: {niter} iterations x {nchains} chains x {len(prob["data"])} points: GPU {tg:.2f} s '
      f'({nchains*niter/tg/1e6:.1f} M/s, '
      f'plan {gpu.plan}, kernel {gpu.last_kernel}), oracle {to:.1f} s on {nthreads} threads; identical: {same}; '
      f'accepted {int(st["numaccept"].sum())}')
sys.exit(0 if same else 1)
