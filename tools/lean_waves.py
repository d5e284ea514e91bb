#!/usr/bin/env python3
"""The hand-laid-out DEMC loop with 1, 2 or 4 waves per chain (lean_demc_kernel /
lean_demc_wide_kernel) over ensemble sizes: million chain-steps/s, C2's data.

    python tools/lean_waves.py [nchains ...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == '--one':
    sys.path.insert(0, ROOT)
    import time
    import bench
    from mc3_amd import engine
    n, nref = int(sys.argv[2]), int(sys.argv[3])
    prob = bench.c2_problem(n, 2*nref + 64)
    with engine.DeviceChains(**prob) as g:
        Z0, lp0 = bench.initial_population(prob, g.eval_chisq)
        g.set_population(Z0, lp0)
        g.run(nref)
        t0 = time.perf_counter()
        g.run(nref)
        dt = time.perf_counter() - t0
        print(json.dumps({'nchains': n, 'waves': g.plan['waves_per_chain'], 'fast': g.plan['fast'],
                          'million_chain_steps_per_s': round(n*nref/dt/1e6, 1),
                          'us_per_iteration': round(dt/nref*1e6, 3)}), flush=True)
    sys.exit(0)

sizes = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 384, 512, 768, 1024, 2048]
for n in sizes:
    for w in (1, 2, 4):
        env = dict(os.environ, MC3HIP_LEAN_WAVES=str(w))
        nref = 40000 if n <= 512 else 16000
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--one', str(n), str(nref)],
                              env=env)
