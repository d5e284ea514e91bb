#!/usr/bin/env python3
"""One-off soak: a long run on the GPU against the oracle (OpenMP on the host cores), compared
exactly.  python tests/soak.py [iterations] [sampler] [nchains] [wavelet ndata]
(C2's shape by default; `wavelet 16384` = C4's)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mc3_amd import engine  # noqa: E402
import helpers as H  # noqa: E402

niter = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
sampler = sys.argv[2] if len(sys.argv) > 2 else 'demc'
nchains = int(sys.argv[3]) if len(sys.argv) > 3 else 256
if len(sys.argv) > 5 and sys.argv[4] == 'wavelet':
    prob = H.wavelet_problem(ndata=int(sys.argv[5]), nchains=nchains, nsamples=nchains*niter,
                             sampler=sampler)
else:
    prob = H.quad_problem(ndata=1000, nchains=nchains, nsamples=nchains*niter, sampler=sampler)
orc = H.make_oracle(prob)
Z0, lp0 = H.initial_population(orc)
orc.set_population(Z0, lp0)
gpu = engine.DeviceChains(**prob)
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
print(f'{sampler}: {niter} iterations x {nchains} chains x {len(prob["data"])} points: GPU {tg:.2f} s '
      f'({nchains*niter/tg/1e6:.1f} M/s, '
      f'plan {gpu.plan}), oracle {to:.1f} s on {nthreads} threads; identical: {same}; '
      f'accepted {int(st["numaccept"].sum())}')
sys.exit(0 if same else 1)
