#!/usr/bin/env python3
"""Iteration time of the tutorial's quadratic across the fused / tiled boundary (data
lengths around 8192..32768, ensembles of 256..2048 chains): which plan the library picks and
what an iteration costs.  MC3HIP_NO_TILED=1 forces the fused kernels for comparison."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import helpers as H  # noqa: E402
import tools_pop  # noqa: E402
from mc3_amd import engine  # noqa: E402

for ndata, nchains in ((8192, 256), (8192, 512), (8192, 768), (8192, 1024), (12000, 512), (12000, 1024),
                       (16000, 512), (16383, 256), (16384, 256), (16000, 2048), (32768, 256)):
    n = 300
    prob = H.quad_problem(ndata=ndata, nchains=nchains, nsamples=nchains*(n + 60))
    g = engine.DeviceChains(**prob)
    Z0, lp0 = tools_pop.population(g)
    g.set_population(Z0, lp0)
    g.run(50)
    g.run(n)
    us = g.last_run_ms()*1e3/n
    pl = g.plan
    print(f'ndata {ndata:6d} chains {nchains:5d}: {us:8.2f} us/iter = {nchains/us:7.2f} M steps/s; '
          f'tile {pl["tile"]} resident {pl["resident"]} fast {pl["fast"]} '
          f'waves/chain {pl["waves_per_chain"]}', flush=True)
    g.close()
