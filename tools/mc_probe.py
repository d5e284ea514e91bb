#!/usr/bin/env python3
"""Development probe: what a turn costs when a wave advances several chains per refresh.

C2-shaped problems (quadratic, ndata 1000, demc) at a few ensemble sizes, with the number
of co-resident workgroups capped (MC3HIP_PERSIST_CAPACITY) so that two or four chains share
a wave: microseconds per refresh.  `MC3HIP_NO_LEAN=1 python tools/mc_probe.py` shows the
general instance (run_chain MC) instead of lean_demc_mc_kernel.  DESIGN.md section 4.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from mc3_amd import engine  # noqa: E402
import helpers as H  # noqa: E402
import bench_configs as B  # noqa: E402

for nch, cap in ((128, 0), (128, 64), (128, 32), (512, 256), (2048, 1024), (1024, 0)):
    if cap:
        os.environ['MC3HIP_PERSIST_CAPACITY'] = str(cap)
    else:
        os.environ.pop('MC3HIP_PERSIST_CAPACITY', None)
    n = 6000
    prob = H.quad_problem(ndata=1000, nchains=nch, nsamples=nch*(n + 700), sampler='demc')
    g = engine.DeviceChains(**prob)
    Z0, lp0 = B.population(g, prob)
    g.set_population(Z0, lp0)
    g.run(600)
    t0 = time.perf_counter()
    g.run(n)
    dt = time.perf_counter() - t0
    print(f"{nch} chains, capacity {cap or 'as is'}: {g.plan['chains_per_workgroup']} per wave, "
          f"fast={g.plan['fast']}, {dt/n*1e6:.3f} us per refresh", flush=True)
    g.close()
