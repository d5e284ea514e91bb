#!/usr/bin/env python3
"""Development aid: where does the ~75 ms wait of repeated sample() calls sit at ENGINE level?
Contexts created, run and (mode 'close') destroyed one after another; per phase wall ms.
    python tools/stall_probe_engine.py [close|keep|reuse]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mc3_amd import engine
import helpers as H
mode = sys.argv[1] if len(sys.argv) > 1 else 'close'
prob = H.linear_problem(ndata=100000, nmodel=10, nchains=512, nsamples=512*200)
keep = []
gpu = None
for i in range(8):
    t = [time.perf_counter()]
    if mode != 'reuse' or gpu is None:
        gpu = engine.DeviceChains(**prob)
    t.append(time.perf_counter())
    if mode != 'reuse' or i == 0:
        Z0, lp0, filled, failed = gpu.init_population()
        gpu.set_population(Z0, lp0)
    t.append(time.perf_counter())
    steps = []
    for k in range(10):
        t0 = time.perf_counter()
        if gpu.refresh_done + 20 > gpu.nzchain:
            gpu.recycle()
        gpu.run(20)
        st = gpu.state()
        steps.append((time.perf_counter() - t0)*1e3)
    t.append(time.perf_counter())
    Z, lp, zc = gpu.rows()
    t.append(time.perf_counter())
    if mode == 'close':
        gpu.close()
    elif mode == 'keep':
        keep.append(gpu)
    t.append(time.perf_counter())
    d = np.diff(t)*1e3
    print(f'{mode} {i}: create {d[0]:6.1f} init {d[1]:6.1f} runs {d[2]:6.1f} (max step {max(steps):5.1f}) '
          f'rows {d[3]:5.1f} close {d[4]:5.1f} total {sum(d):6.1f} ms', flush=True)
