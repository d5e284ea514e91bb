#!/usr/bin/env python3
"""Development tool: persistent-kernel rate of the three samplers on the C2 workload."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from mc3_amd import engine
for smp in ('mrw', 'demc', 'snooker'):
    prob = bench.c2_problem(256, 22000)
    prob['sampler'] = smp
    g = engine.DeviceChains(**prob)
    Z0, lp0 = bench.initial_population(prob, g.eval_chisq)
    g.set_population(Z0, lp0)
    g.run(2000)
    t0 = time.perf_counter(); g.run(20000); dt = time.perf_counter() - t0
    print(smp, g.plan, round(256*20000/dt/1e6, 1), 'M/s', round(dt/20000*1e6, 3), 'us/iter')
    g.close()
