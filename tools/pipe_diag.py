#!/usr/bin/env python3
"""Diagnostic: where a slot of the pipelined wavelet kernel (wavelet_pipe_kernel) goes.

Builds a SEPARATE library with -DMC3HIP_PIPE_DIAG (mc3_amd/libmc3hip_pipediag.so) in which the
chain wave and the tail wave of workgroup 0 print, per segment of a slot -- (1) -> (A) -> first
-> second -> third -> fourth halo barrier -> (1) -- the nanoseconds (the 100 MHz clock: s_memtime stalls under FP64 load) they work and the
segment lasts; a segment that lasts longer than both work is bound by the compute waves.

    python tools/pipe_diag.py [c4x2|c4x2wide|c4x2mrw|c4x4]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from mc3_amd import build as b, _lib  # noqa: E402

# (PIPE_DIAG_EXTRA="-DX -DY": further macros, into a library of their own name PIPE_DIAG_LIB)
lib = os.path.join(ROOT, 'mc3_amd', os.environ.get('PIPE_DIAG_LIB', 'libmc3hip_pipediag.so'))
if '--build' in sys.argv or not os.path.exists(lib):
    b.build(force=True, verbose=False, lib=lib,
            extra=['-DMC3HIP_DEV', '-DMC3HIP_PIPE_DIAG'] + os.environ.get('PIPE_DIAG_EXTRA', '').split())
    if '--build' in sys.argv:
        sys.exit(0)
_lib.LIB_PATH = lib
from mc3_amd import engine  # noqa: E402
import bench_configs as BC  # noqa: E402

name = ([a for a in sys.argv[1:] if not a.startswith('--')] or ['c4x2wide'])[0]
N = int(os.environ.get('PIPE_DIAG_ITERS', '600'))
prob, kw = BC.CONFIGS[name](N)
g = engine.DeviceChains(**prob, **kw)
Z0, lp0 = BC.population(g, prob)
g.set_population(Z0, lp0)
g.run(N)
g.state()
print(g.plan)
g.close()
