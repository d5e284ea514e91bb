import os, sys, time, gc
import numpy as np
ROOT='/root/repo' if os.path.exists('/root/repo/mc3_amd') else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import mc3_amd
from mc3_amd import utils as mu
import helpers as H
mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'
prob = H.linear_problem(ndata=100000, nmodel=10)
kw = dict(data=prob['data'], uncert=prob['uncert'], func=lambda p, A: p @ A,
          params=prob['params'], indparams=[prob['indep']], pstep=prob['pstep'],
          pmin=prob['pmin'], pmax=prob['pmax'], prior=prob['prior'],
          priorlow=prob['priorlow'], priorup=prob['priorup'],
          sampler='demc', nsamples=1e5, burnin=100, nchains=512,
          grtest=True, log=mu.Log(verb=0), seed=3)
keep = []
if mode == 'nogc':
    gc.disable()
for i in range(6):
    if mode == 'sleep' and i:
        time.sleep(0.3)
    t0 = time.perf_counter()
    out = mc3_amd.sample(**kw)
    dt = time.perf_counter() - t0
    if mode in ('keep', 'nogc'):
        keep.append(out)
    print(f'{mode} call {i}: {dt*1e3:7.1f} ms', flush=True)
