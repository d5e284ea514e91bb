import sys, os, json
import numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests'); sys.path.insert(0, ROOT+'/tools')
import helpers as H, tools_pop
from mc3_amd import engine
def measure(prob, n):
    g = engine.DeviceChains(**prob)
    Z0, lp0 = tools_pop.population(g); g.set_population(Z0, lp0)
    g.run(n//5); g.run(n)
    us = g.last_run_ms()*1e3/(n*g.thinning); pl = g.plan; g.close()
    return us, pl
PRIOR = dict(prior=np.array([3.0, 0.0, 0.5]), priorlow=np.array([0.5, 0.0, 0.02]), priorup=np.array([0.25, 0.0, 0.04]))
for smp in ('demc','snooker','mrw'):
    for nd in (1000, 2048, 4096, 8192, 16384):
        for name, kw in (('plain', {}), ('prior', PRIOR), ('fixed', 'fixed'), ('eps', dict(fepsilon=0.01))):
            if name=='eps' and smp=='mrw': continue
            n = 3000
            prob = H.quad_problem(ndata=nd, nchains=256, nsamples=256*(n+n//5+8), sampler=smp, **(kw if isinstance(kw, dict) else {}))
            if kw == 'fixed':
                ps = prob['pstep'].copy(); ps[1] = 0.0; prob['pstep'] = ps
            try:
                us, pl = measure(prob, n)
                print(json.dumps(dict(sampler=smp, ndata=nd, shape=name, us=round(us,3), Msteps=round(256/us,1), fast=pl['fast'], wpc=pl['waves_per_chain'], tile=pl['tile'], resident=pl['resident'])), flush=True)
            except Exception as e:
                print(json.dumps(dict(sampler=smp, ndata=nd, shape=name, error=str(e)[:100])), flush=True)
