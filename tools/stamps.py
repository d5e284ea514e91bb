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
from mc3_amd import build as b, _lib  # noqa: E402

lib = os.path.join(ROOT, 'mc3_amd', 'libmc3hip_stamps.so')
if '--build' in sys.argv or not os.path.exists(lib):
    subprocess.check_call([b.hipcc()] + b.FLAGS + ['-DMC3HIP_STAMPS', '-o', lib, b.SRC])
    if '--build' in sys.argv:
        sys.exit(0)
_lib.LIB_PATH = lib
from mc3_amd import engine  # noqa: E402
import bench  # noqa: E402

wpc = os.environ.get('MC3HIP_WPC', '1')
prob = bench.c2_problem(256, 3000)
g = engine.DeviceChains(no_graph=True, **prob)
Z0, lp0 = bench.initial_population(prob, g.eval_chisq)
g.set_population(Z0, lp0)
g.run(500)
names = ['entry->loads', 'loads->proposal', 'proposal->chisq', 'chisq->end']
acc = np.zeros(4)
n = 0
for _ in range(200):
    g.run(1)
    out = (ctypes.c_uint64*16)()
    _lib.check(g.lib.mc3hip_debug_stamps(g._ctx, out))
    t = np.array(out[:5], dtype=np.float64)
    if t[3] == 0 or np.any(np.diff(t) < 0):
        continue            # out-of-bounds proposal: no chisq phase
    acc += np.diff(t)
    n += 1
print(f'WPC={wpc}  {n} launches, mean shader-clock cycles per phase (workgroup 0):')
for k, v in zip(names, acc/n):
    print(f'  {k:18s} {v:9.0f}')
print(f'  {"total":18s} {acc.sum()/n:9.0f}')
