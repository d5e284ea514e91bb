#!/usr/bin/env python3
"""Diagnostic (separate -DMC3HIP_STAMPS build): average shader-clock cycles per phase
of one iteration of the persistent kernel (workgroup 0, non-draining ticks)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mc3_amd import _lib  # noqa: E402
_lib.LIB_PATH = (os.environ.get('MC3HIP_TICKS_LIB')
                 or os.path.join(ROOT, 'mc3_amd', 'libmc3hip_stamps.so'))
from mc3_amd import engine  # noqa: E402
import bench  # noqa: E402

# python tools/ticks.py [thinning]          C2 (lean kernels)
# python tools/ticks.py c4                  C4's data with sampler='demc' (the general
#                                           persistent kernel keeps tick sums for demc only)
if len(sys.argv) > 1 and sys.argv[1] in ('c5', 'q5'):
    # python tools/ticks.py c5 | q5     the resident tiled plans: phases of chain 0's owner wave
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import helpers as H
    import tools_pop
    NIT = 400
    if sys.argv[1] == 'q5':     # the polynomial flavour of the resident plan
        nd = int(sys.argv[2]) if len(sys.argv) > 2 else 100000      # [ndata [nchains]]
        nc = int(sys.argv[3]) if len(sys.argv) > 3 else 512
        prob = H.quad_problem(ndata=nd, nchains=nc, nsamples=nc*(NIT + 60))
    else:
        prob = H.linear_problem(ndata=100000, nchains=512, nsamples=512*(NIT + 60))
    g = engine.DeviceChains(**prob)
    Z0, lp0 = tools_pop.population(g)
    g.set_population(Z0, lp0)
    g.run(50)

    def stamps(page):
        os.environ['MC3HIP_STAMP_PAGE'] = str(page)
        out = (ctypes.c_uint64*16)()
        _lib.check(g.lib.mc3hip_debug_stamps(g._ctx, out))
        return np.array(out[:], dtype=np.float64)
    a0, a1 = stamps(0), stamps(1)
    g.run(NIT)
    b0, b1 = stamps(0) - a0, stamps(1) - a1
    n = b0[12]
    print(f'plan={g.plan}; iterations sampled (in bounds, chain 0): {int(n)}; '
          f'wall {g.last_run_ms()*1e3/NIT:.2f} us per iteration')
    rows = [('top -> partner rows arrived', b0[8]), ('-> proposal published', b1[0]),
            ('-> tile phase done (incl. waiting for the column\'s proposals)', b1[1]),
            ('-> all tile sums collected', b1[2]), ('-> chisq known', b1[3]),
            ('-> end of iteration (accept)', b0[10]), ('-> row published', b0[11]),
            ('-> top of next iteration', b0[13])]
    for k, v in rows:
        print(f'  {k:62s} {v/n:9.0f} cycles')
    print(f'  {"sum":62s} {sum(v for _, v in rows)/n:9.0f}')
    if g.plan.get('owner_waves', 0) > 0 and b1[4] > 0:
        # resident_async_kernel: tile wave 0 of workgroup 0 (per iteration of the launch)
        # (on the 100 MHz clock: the shader clock's count stalls under FP64 matrix instructions)
        u = lambda v: v/NIT/100.0                       # noqa: E731
        print(f'  tile wave 0 of workgroup 0: blocks multiplied {b1[4]/NIT:.2f} per iteration, '
              f'turns with nothing in hand {b1[5]/NIT:.2f}; us per iteration: products (with the '
              f'looks for the next block) {u(b1[6]):.2f}, waiting for rows {u(b1[7]):.2f}, row '
              f'totals left and counted {u(b1[9]):.2f}, chains finished {u(b1[10]):.2f}, from '
              f'there to the next block\'s products {u(b1[11]):.2f}, in all {u(b1[8]):.2f}')
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == 'c4cw':
    # python tools/ticks.py c4cw [sampler] [wide]    C4 (wide: bounds that reject nothing) with the chain-wave kernel: phases of the chain
    # wave of workgroup 0 (kernels_wavelet.h)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import helpers as H
    import tools_pop
    NIT = 4000
    smp = sys.argv[2] if len(sys.argv) > 2 else 'snooker'
    wide = dict(pmin=np.full(6, -1e3), pmax=np.full(6, 1e3)) if 'wide' in sys.argv else {}
    prob = H.wavelet_problem(ndata=16384, nchains=256, nsamples=256*(NIT + 600), sampler=smp,
                             **wide)
    g = engine.DeviceChains(**prob)
    Z0, lp0 = tools_pop.population(g)
    g.set_population(Z0, lp0)
    g.run(500)

    def stamps(page):
        os.environ['MC3HIP_STAMP_PAGE'] = str(page)
        out = (ctypes.c_uint64*16)()
        _lib.check(g.lib.mc3hip_debug_stamps(g._ctx, out))
        return np.array(out[:], dtype=np.float64)
    a0, a1 = stamps(1), stamps(1)       # (the chain wave's sums are all on the second page)
    g.run(NIT)
    b0, b1 = stamps(1) - a0, stamps(1) - a1
    n = b0[12]
    print(f'{smp}: plan={g.plan}; iterations sampled (in bounds, chain 0): {int(n)}; '
          f'wall {g.last_run_ms()*1e3/NIT:.2f} us per iteration')
    rows = [('top -> jump known (rows arrived)', b0[8]), ('-> proposal published', b1[0]),
            ('-> likelihood done (the compute waves)', b1[1]),
            ('-> result collected', b1[2]), ('-> chisq known (priors)', b1[3]),
            ('-> end of iteration (accept)', b0[10]), ('-> row published', b0[11]),
            ('-> top of next iteration', b0[13])]
    for k, v in rows:
        print(f'  {k:62s} {v/n:9.0f} cycles')
    print(f'  {"sum":62s} {sum(v for _, v in rows)/n:9.0f}')
    if g.plan['fast'] == 3:
        print(f'  (inside "likelihood": the chain wave\'s preparation of the next iteration '
              f'{b0[14]/n:.0f} cycles)')
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == 'c4':
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import helpers as H
    import tools_pop
    thin = 1
    prob = H.wavelet_problem(ndata=16384, nchains=256, nsamples=256*20000, sampler='demc')
    g = engine.DeviceChains(**prob)
    Z0, lp0 = tools_pop.population(g)
else:
    thin = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    prob = bench.c2_problem(256, 20000, thin)
    g = engine.DeviceChains(**prob)
    Z0, lp0 = bench.initial_population(prob, g.eval_chisq)
g.set_population(Z0, lp0)
g.run(2000)
out = (ctypes.c_uint64*16)()
_lib.check(g.lib.mc3hip_debug_stamps(g._ctx, out))
a = np.array(out[:], dtype=np.float64)
g.run(16000)
_lib.check(g.lib.mc3hip_debug_stamps(g._ctx, out))
b = np.array(out[:], dtype=np.float64) - a
n = b[12]
print(f'thinning={thin} plan={g.plan} iterations sampled (in-bounds, workgroup 0): {int(n)}')
print(f'  top -> partner rows arrived   {b[8]/n:8.0f} cycles')
print(f'     (top -> loads issued {b[15]/n:6.0f}; extra polls per iteration {b[14]/n:.3f})')
print(f'  -> chisq reduced              {b[9]/n:8.0f}')
print(f'  -> end of iteration           {b[10]/n:8.0f}')
print(f'  -> row published              {b[11]/n:8.0f}')
print(f'  -> top of next iteration      {b[13]/n:8.0f}')
print(f'  sum                           {(b[8]+b[9]+b[10]+b[11]+b[13])/n:8.0f}   '
      f'(wall {g.last_run_ms()*1e3/16000/thin:.3f} us per iteration)')
