#!/usr/bin/env python3
"""Test-side check (uses the CPU checker, hence under tests/): a fixed polynomial coefficient on the lean persistent instances, against the oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import bench
from mc3_amd import engine
import oracle_engine
for smp in ('demc', 'snooker', 'mrw'):
    prob = bench.c2_problem(256, 3000)
    prob['sampler'] = smp
    prob['pstep'] = np.array([1.0, 0.0, 0.1])      # the linear coefficient is fixed
    prob['params'] = np.array([10.0, -2.4, 0.1])
    g = engine.DeviceChains(**prob)
    o = oracle_engine.OracleChains(**prob)
    rs = np.random.RandomState(5)
    M0 = prob['hsize']*prob['nchains']
    Z0 = np.column_stack([rs.normal(3.0, 0.3, M0), rs.normal(0.5, 0.01, M0)])
    full = np.tile(prob['params'], (M0, 1))
    full[:, [0, 2]] = Z0
    lp0 = -0.5*g.eval_chisq(full)
    g.set_population(Z0, lp0); o.set_population(Z0, lp0)
    g.run(2500); o.run(2500)
    nrow = g.M0 + 2500*prob['nchains']
    Zg, lg, zg = g.rows(0, nrow); Zo, lo, zo = o.rows(0, nrow)
    same = np.array_equal(Zg, Zo) and np.array_equal(lg, lo) and np.array_equal(g.state()['numaccept'], o.state()['numaccept'])
    t0 = time.perf_counter(); g.run(400); dt = time.perf_counter() - t0
    print(f'{256*400/dt/1e6:.1f} M chain-steps/s', end=' ')
    print(smp, 'plan', g.plan['fast'], g.plan['persistent'], 'identical to oracle:', same, 'accepted', int(g.state()['numaccept'].sum()))
    g.close()
