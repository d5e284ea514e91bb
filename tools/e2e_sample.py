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
config = sys.argv[4] if len(sys.argv) > 4 else 'c2'
if config == 'c5':
    # python tools/e2e_sample.py 1e6 512 1 c5: the 10-coefficient design matrix x 1e5 points
    prob = H.linear_problem(ndata=100000, nmodel=10)
    kw = dict(data=prob['data'], uncert=prob['uncert'], func=lambda p, A: p @ A,
              params=prob['params'], indparams=[prob['indep']], pstep=prob['pstep'],
              pmin=prob['pmin'], pmax=prob['pmax'], prior=prob['prior'],
              priorlow=prob['priorlow'], priorup=prob['priorup'],
              sampler='demc', nsamples=nsamples, burnin=100, nchains=nchains,
              grtest=grtest, log=mu.Log(verb=0), seed=3)
else:
    x, data, uncert = H.tutorial_data(1000)
    kw = dict(data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
              indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
              pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
              sampler='demc', nsamples=nsamples, burnin=100, nchains=nchains,
              grtest=grtest, log=mu.Log(verb=0), seed=3)
from mc3_amd import engine as _engine  # noqa: E402


class Reporting(_engine.DeviceChains):
    """DeviceChains that tells which launch structure the library chose and how long the
    device worked."""
    log = []
    syncs = []
    runs = []

    def run(self, *a, **k):
        t0 = time.perf_counter()
        out = super().run(*a, **k)
        Reporting.runs.append(round((time.perf_counter() - t0)*1e3, 2))
        Reporting.plan = self.plan
        return out

    def state(self):
        t0 = time.perf_counter()
        super().sync()
        tsync = (time.perf_counter() - t0)*1e3
        out = super().state()
        dt = (time.perf_counter() - t0)*1e3
        Reporting.syncs.append(round(tsync, 2))
        try:
            Reporting.log.append((out['refresh_done'], round(dt, 2), round(self.last_run_ms(), 2)))
        except Exception:
            pass
        return out


kw['engine_factory'] = Reporting
np.random.seed(1)
mc3_amd.sample(**dict(kw, nsamples=nchains*200))       # warm up (library load, JIT-free)
Reporting.log = []
Reporting.syncs = []
Reporting.runs = []
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
print('plan:', Reporting.plan)
print('state() calls (refresh_done, wall ms, last run ms):', Reporting.log)
print('  of which sync:', Reporting.syncs, ' run() calls:', Reporting.runs)
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
