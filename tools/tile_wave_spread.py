#!/usr/bin/env python3
"""Diagnostic (the -DMC3HIP_STAMPS build, tools/stamps.py): resident_async_kernel's tile waves
one by one for C5 -- time waiting for rows, turns with nothing in hand, time in the products --
summarised by the wave's place in its workgroup (waves v and v + 4 share SIMD v; SIMDs 0..2
carry an owner wave, SIMD 3 of the workgroups of tile 0 the column's scheduler).
    python tools/stamps.py build; python tools/tile_wave_spread.py"""
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
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import helpers as H  # noqa: E402
import tools_pop  # noqa: E402

NIT = 400
prob = H.linear_problem(ndata=100000, nchains=512, nsamples=512*(NIT + 60))
g = engine.DeviceChains(**prob)
Z0, lp0 = tools_pop.population(g)
g.set_population(Z0, lp0)
g.run(50)
ntiles, ncols = 49, 5


def table():
    out = (ctypes.c_uint64*16)()
    rows = []
    for page in range(256, 256 + (ntiles*ncols*8*4 + 15)//16):
        os.environ['MC3HIP_STAMP_PAGE'] = str(page)
        _lib.check(g.lib.mc3hip_debug_stamps(g._ctx, out))
        rows.append(np.array(out[:], dtype=np.float64))
    return np.concatenate(rows)[:ntiles*ncols*8*4].reshape(ncols, ntiles, 8, 4)


a = table()
g.run(NIT)
b = (table() - a)/NIT
print(f'plan={g.plan}; wall {g.last_run_ms()*1e3/NIT:.2f} us per iteration')
us = 0.01       # a tick of the 100 MHz clock
print('per iteration, by wave of the workgroup (mean over the 245 workgroups; min .. max):')
for wv in range(8):
    w, idle, mul, tot = (b[:, :, wv, k] for k in range(4))
    print(f'  wave {wv} (SIMD {wv % 4}): waiting for rows {w.mean()*us:5.2f} us '
          f'({w.min()*us:5.2f} .. {w.max()*us:5.2f}), turns with nothing in hand '
          f'{idle.mean():5.2f} ({idle.min():4.2f} .. {idle.max():4.2f}), products '
          f'{mul.mean()*us:5.2f} us ({mul.min()*us:5.2f} .. {mul.max()*us:5.2f})')
print('by tile (waves 0..7 together): tile, waiting for rows, turns with nothing in hand')
for t in (0, 1, 2, 24, 47, 48):
    print(f'  tile {t:2d}: {b[:, t, :, 0].mean()*us:5.2f} us  {b[:, t, :, 1].mean():5.2f}')
slow = np.unravel_index(np.argmin(b[:, :, :, 0] + 100*b[:, :, :, 1]), b.shape[:3])
print(f'the wave that waited least: column {slow[0]} tile {slow[1]} wave {slow[2]}: '
      f'rows {b[slow][0]*us:.2f} us, idle turns {b[slow][1]:.2f}, products {b[slow][2]*us:.2f} us')
print('by column of chains (all tiles, all waves): products us, waiting for rows us, idle turns, '
      'in all us')
for c in range(ncols):
    print(f'  column {c}: {b[c, :, :, 2].mean()*us:6.2f} {b[c, :, :, 0].mean()*us:6.2f} '
          f'{b[c, :, :, 1].mean():6.2f} {b[c, :, :, 3].mean()*us:7.2f}')
print('products us by tile (rows) and column (wave 0 | wave 4):')
for t in range(0, ntiles, 4):
    print(f'  tile {t:2d}: ' + ' '.join(f'{b[c, t, 0, 2]*us:5.1f}|{b[c, t, 4, 2]*us:5.1f}'
                                        for c in range(ncols)))
