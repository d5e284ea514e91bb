#!/usr/bin/env python3
"""End-to-end wall time of mc3_amd.sample() (host logic + device) on the
tutorial-style quadratic fit, with a breakdown of where the host spends time."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import mc3_amd  # noqa: E402
from mc3_amd import utils as mu  # noqa: E402
import helpers as H  # noqa: E402

nsamples = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
nchains = int(sys.argv[2]) if len(sys.argv) > 2 else 256
grtest = (sys.argv[3] != '0') if len(sys.argv) > 3 else True
x, data, uncert = H.tutorial_data(1000)
kw = dict(data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
          indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
          pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
          sampler='demc', nsamples=nsamples, burnin=100, nchains=nchains,
          grtest=grtest, log=mu.Log(verb=0), seed=3)
np.random.seed(1)
mc3_amd.sample(**dict(kw, nsamples=nchains*200))       # warm up (library load, JIT-free)
np.random.seed(1)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
out = mc3_amd.sample(**kw)
pr.disable()
dt = time.perf_counter() - t0
n = len(out['posterior'])
print(f'sample(): {n} samples, {nchains} chains, grtest={grtest}: {dt:.3f} s wall '
      f'= {n/dt/1e6:.2f} M samples/s end to end; acceptance {out["acceptance_rate"]:.1f}%')
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
