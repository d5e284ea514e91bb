#!/usr/bin/env python3
"""Development tool: persistent-kernel rate of the quadratic fit as ndata grows (256 chains, demc)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mc3_amd import engine  # noqa: E402

SAMPLER = sys.argv[1] if len(sys.argv) > 1 else "demc"
for nd in ((1000, 1024, 2000, 4000, 8000, 16000) if len(sys.argv) < 3 else (1000, 2000, 4000)):
    bench.NDATA = nd
    prob = bench.c2_problem(256, 6000)
    prob['sampler'] = SAMPLER
    g = engine.DeviceChains(**prob)
    Z0, lp0 = bench.initial_population(prob, g.eval_chisq)
    g.set_population(Z0, lp0)
    g.run(1000)
    t0 = time.perf_counter()
    g.run(4000)
    dt = time.perf_counter() - t0
    print(SAMPLER, nd, g.plan, round(256*4000/dt/1e6, 1), 'M chain-steps/s', round(dt/4000*1e6, 2), 'us/iter')
    g.close()
