#!/usr/bin/env python3
"""Diagnostic: where one step-kernel launch spends its cycles.

Builds a SEPARATE library with -DMC3HIP_STAMPS (mc3_amd/libmc3hip_stamps.so; the
product library never contains the stamps), runs the C2 workload eagerly and
prints the shader-clock deltas between the phase stamps of workgroup 0.  Each
stamp first drains outstanding memory operations, so the build measures phase
SHARES, not the product kernel's run time.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from mc3_amd import build as b, _lib  # noqa: E402

lib = os.path.join(ROOT, 'mc3_amd', 'libmc3hip_stamps.so')
if '--build' in sys.argv or not os.path.exists(lib):
    # The LDS stamps of the wavelet path (MC3_LSTAMP) are off by default: with them in, this
    # compiler version fails the whole translation unit ("illegal VGPR to SGPR copy" at the
    # scalar pins of the lean kernels); `--lstamp` tries anyway.  Without them `c4` reports
    # only the three coarse phases; DESIGN.md's C4 breakdown comes from ablation builds.
    extra = [] if '--lstamp' in sys.argv else ['-DMC3HIP_NO_LSTAMP']
    b.build(force=True, verbose=True, lib=lib, extra=['-DMC3HIP_DEV', '-DMC3HIP_STAMPS'] + extra)
    if '--build' in sys.argv:
        sys.exit(0)
_lib.LIB_PATH = lib
from mc3_amd import engine  # noqa: E402
import bench  # noqa: E402

import helpers as H  # noqa: E402

which = ([a for a in sys.argv[1:] if not a.startswith('--')] or ['c2'])[0]
if which == 'c4':
    # `wide`: bounds that reject nothing (C4's own reject 3 in 10 snooker proposals before the
    # likelihood); `demc` / `mrw`: the other samplers
    smp = ([a for a in sys.argv[2:] if a in ('demc', 'mrw', 'snooker')] or ['snooker'])[0]
    wide = dict(pmin=np.full(6, -1e3), pmax=np.full(6, 1e3)) if 'wide' in sys.argv else {}
    prob = H.wavelet_problem(ndata=16384, nchains=256, nsamples=256*1200, sampler=smp, **wide)
    ini = H.initial_population
else:
    prob = bench.c2_problem(256, 3000)
mode = os.environ.get('STAMP_MODE', 'persist')
kw = dict(no_graph=True, no_persist=True) if mode == 'launch' else {}
g = engine.DeviceChains(**prob, **kw)
if which == 'c4':
    import tools_pop
    Z0, lp0 = tools_pop.population(g)
else:
    Z0, lp0 = bench.initial_population(prob, g.eval_chisq)
g.set_population(Z0, lp0)
g.run(100)
names = ['5 top -> 2 proposal done', '2 -> 3 chisq done', '3 -> 6 iteration end']
acc = np.zeros(3)
n = 0
for _ in range(60):
    g.run(5 if mode != 'launch' else 1)
    out = (ctypes.c_uint64*16)()
    _lib.check(g.lib.mc3hip_debug_stamps(g._ctx, out))
    t = np.array([out[5], out[2], out[3], out[6]], dtype=np.float64)
    if np.any(np.diff(t) <= 0):
        continue            # out-of-bounds proposal in the last iteration: no chisq phase
    acc += np.diff(t)
    n += 1
    if which == 'c4':
        sel = (5, 2, 8, 14, 9, 10, 12, 13, 3, 6)
        wl_acc = wl_acc + np.diff([float(out[i]) for i in sel]) if n > 1 \
            else np.diff([float(out[i]) for i in sel])
print(f'{which} mode={mode} plan={g.plan}: {n} samples, mean shader-clock cycles per phase '
      '(workgroup 0, last iteration of a launch):')
for k, v in zip(names, acc/max(n, 1)):
    print(f'  {k:28s} {v:9.0f}')
print(f'  {"iteration":28s} {acc.sum()/max(n, 1):9.0f}')
if n == 0:
    print('  (no samples: the hand-laid-out DEMC loop keeps per-phase tick sums instead -- '
          'tools/ticks.py; or STAMP_MODE=launch / MC3HIP_NO_LEAN=1 for the stamped kernels)')
if which == 'c4' and g.plan['waves_per_chain'] == 8 and not os.environ.get('MC3HIP_NO_CHAINWAVE'):
    # the chain-wave kernel: compute wave 0's stamps of the last iteration
    sel = (0, 8, 14, 9, 10, 12, 13, 1, 2)
    lab = ['(A) proposal there -> pyramid entry', '-> level 0 done', '-> levels 1.. done (stage A)',
           '-> barrier (1) + stage B', '-> stage C', '-> sums, chisq', '-> result written',
           '-> barrier (3) passed (waits for the chain wave)']
    acc2 = np.zeros(len(lab))
    m = 0
    for _ in range(60):
        g.run(7)
        out = (ctypes.c_uint64*16)()
        _lib.check(g.lib.mc3hip_debug_stamps(g._ctx, out))
        d = np.diff([float(out[i]) for i in sel])
        if np.all(d > 0) and d.sum() < 1e6:
            acc2 += d
            m += 1
    print(f'  chain-wave kernel, compute wave 0, {m} samples:')
    for k, v in zip(lab, acc2/max(m, 1)):
        print(f'    {k:50s} {v:9.0f}')
    print(f'    {"sum (A) -> (3)":50s} {acc2.sum()/max(m, 1):9.0f}')
elif which == 'c4':
    lab = ['top -> proposal known (barrier)', '-> wavelet entry',
           '-> level 0 done (wave 0)', '-> levels 1.. done (stage A)',
           '-> barrier 1 + stage B (wave 0)', '-> stage C (wave 0)',
           '-> barriers 2, 3 + weights + tree', '-> chisq done', '-> iteration end']
    print('  wavelet, mean over the same samples:')
    for k, v in zip(lab, wl_acc/max(n, 1)):
        print(f'    {k:34s} {v:9.0f}')
