#!/usr/bin/env python3
"""BASELINE.json configs[0] on the GPU: the tutorial's quadratic fit (examples/tutorial.py:
100 points, 7 chains, nsamples = 1e5, burnin 1000) through mc3_amd.sample() with each sampler;
wall seconds of the second call (the first pays context creation)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import mc3_amd  # noqa: E402
from mc3_amd import utils as mu  # noqa: E402
import helpers as H  # noqa: E402

x, data, uncert = H.tutorial_data(100)
for smp in ('snooker', 'demc', 'mrw'):
    kw = dict(data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
              indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
              pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
              sampler=smp, nsamples=1e5, burnin=1000, nchains=7, ncpu=7,
              grtest=True, log=mu.Log(verb=0), seed=3)
    mc3_amd.sample(**dict(kw, nsamples=2e4))
    t0 = time.perf_counter()
    out = mc3_amd.sample(**kw)
    dt = time.perf_counter() - t0
    print(f"{smp}: wall {dt:.3f} s, acceptance {out['acceptance_rate']:.1f} %, bestp "
          f"{np.round(out['bestp'], 3)}, red_chisq {out['red_chisq']:.3f}, posterior "
          f"{out['posterior'].shape}", flush=True)
