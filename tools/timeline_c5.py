#!/usr/bin/env python3
"""Diagnostic: the timeline of ONE iteration of the resident tiled plan with the owners as waves
of their own (resident_async_kernel), from a library built with
    python tools/variant_lib.py tl k_resident.hip -DMC3HIP_STAMPS -DMC3HIP_TICKS_ONLY -DMC3HIP_TIMELINE=200
Who did what when (100 MHz clock shared by all CUs; printed in microseconds from the first
proposal of the iteration): per column and block, when the scheduler published it and when the
column's tile waves took / began / left it; per block, when its chains were proposed, collected
and published.

    MC3HIP_TICKS_LIB=mc3_amd/libmc3hip_tl.so python tools/timeline_c5.py
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mc3_amd import _lib  # noqa: E402
_lib.LIB_PATH = (os.environ.get('MC3HIP_TICKS_LIB')
                 or os.path.join(ROOT, 'mc3_amd', 'libmc3hip_tl.so'))
from mc3_amd import engine  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import helpers as H  # noqa: E402
import tools_pop  # noqa: E402

NIT = 300
prob = H.linear_problem(ndata=100000, nchains=512, nsamples=512*(NIT + 60))
g = engine.DeviceChains(**prob)
Z0, lp0 = tools_pop.population(g)
g.set_population(Z0, lp0)
g.run(50)
g.run(NIT)


def pages(first, count):
    out = np.zeros(16*count)
    buf = (ctypes.c_uint64*16)()
    for p in range(count):
        os.environ['MC3HIP_STAMP_PAGE'] = str(first + p)
        _lib.check(g.lib.mc3hip_debug_stamps(g._ctx, buf))
        out[16*p:16*p + 16] = np.array(buf[:], dtype=np.float64)
    return out


print(f'plan={g.plan}; wall {g.last_run_ms()*1e3/NIT:.2f} us per iteration')
own = pages(1024//16, 512*4//16).reshape(512, 4)        # partners, proposed, collected, row
sched = pages(4096//16, 5).reshape(5, 16)
tile = pages(8192//16, 245*32//16).reshape(245, 8, 4)   # taken, rows there, begun, left
blk = pages(16384//16, 512//16).astype(int)
t0 = own[:, 1][own[:, 1] > 0].min()
us = lambda t: (t - t0)/100.0                           # noqa: E731
ntiles = 49
cb = 112
print('column block | scheduler published | tile waves (wave 0 of the 49 workgroups): taken '
      'min/max, products begun min/med/max, sums left min/med/max | chains: proposed min/max, '
      'collected min/med/max, row published med/max')
for col in range(5):
    for b in range(8):
        if sched[col, b] == 0:
            continue
        wgs = [col*ntiles + t for t in range(ntiles)]
        tk = tile[wgs, b, 0]; bg = tile[wgs, b, 2]; lf = tile[wgs, b, 3]
        ch = [c for c in range(col*cb, min(512, (col + 1)*cb)) if blk[c] == b]
        o = own[ch]
        print(f'{col} {b} | {us(sched[col, b]):7.2f} | taken {us(tk.min()):6.2f} {us(tk.max()):6.2f}'
              f'  begun {us(bg.min()):6.2f} {us(np.median(bg)):6.2f} {us(bg.max()):6.2f}'
              f'  left {us(lf.min()):6.2f} {us(np.median(lf)):6.2f} {us(lf.max()):6.2f}'
              f' | {len(ch):2d} chains: partners {us(o[:, 0].min()):6.2f} {us(o[:, 0].max()):6.2f}'
              f' proposed {us(o[:, 1].min()):6.2f} {us(o[:, 1].max()):6.2f}'
              f'  collected {us(o[:, 2].min()):6.2f} {us(np.median(o[:, 2])):6.2f} {us(o[:, 2].max()):6.2f}'
              f'  row {us(np.median(o[:, 3])):6.2f} {us(o[:, 3].max()):6.2f}')
d = own[:, 2] - own[:, 1]
print(f'proposal published -> sums collected: median {np.median(d)/100:.2f} us, max {d.max()/100:.2f}')
d = own[:, 3] - own[:, 2]
print(f'sums collected -> row published: median {np.median(d)/100:.2f} us, max {d.max()/100:.2f}')
d = own[:, 1] - own[:, 0]
print(f'partner rows there -> proposal published: median {np.median(d)/100:.2f} us, max {d.max()/100:.2f}')
print(f'iteration span: first proposal {us(own[:, 1].min()):.2f} .. last row {us(own[:, 3].max()):.2f} us')
g.close()
