import os, sys, json, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools')
import numpy as np
from mc3_amd import engine
import helpers as H
import bench_configs as B
for nch, cap in ((128, 0), (128, 64), (128, 32), (512, 256), (2048, 1024), (1024, 0)):
    if cap: os.environ['MC3HIP_PERSIST_CAPACITY'] = str(cap)
    else: os.environ.pop('MC3HIP_PERSIST_CAPACITY', None)
    n = 6000
    prob = H.quad_problem(ndata=1000, nchains=nch, nsamples=nch*(n+700), sampler='demc')
    g = engine.DeviceChains(**prob)
    Z0, lp0 = B.population(g, prob)
    g.set_population(Z0, lp0)
    g.run(600)
    t0 = time.perf_counter(); g.run(n); dt = time.perf_counter() - t0
    print(nch, cap, g.plan['chains_per_workgroup'], g.plan['fast'], round(dt/n*1e6, 3), 'us per refresh', flush=True)
    g.close()
