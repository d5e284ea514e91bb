#!/usr/bin/env python3
"""A variant of libmc3hip.so that differs from the product in ONE translation unit compiled
with extra flags (measurements: priorities, diagnostic macros) -- the other objects are the
product's own (build/obj), so a variant costs one compilation and a link.

    python tools/variant_lib.py NAME UNIT.hip -DFLAG=1 ...   ->  mc3_amd/libmc3hip_NAME.so
    MC3HIP_LIBRARY=mc3_amd/libmc3hip_NAME.so python tools/bench_configs.py c4
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mc3_amd import build as b  # noqa: E402


def main():
    name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    b.build()                                   # (the product's objects are current)
    objdir = os.path.normpath(b.OBJDIR)
    vobj = os.path.join(objdir, f'variant_{name}_{os.path.splitext(unit)[0]}.o')
    subprocess.check_call([b.hipcc()] + b.CFLAGS + ['-DMC3HIP_DEV'] + flags + ['-c', os.path.join(b.CSRC, unit),
                                                           '-o', vobj])
    objs = [vobj if u == unit else os.path.join(objdir, os.path.splitext(u)[0] + '.o')
            for u in b.UNITS]
    lib = os.path.join(ROOT, 'mc3_amd', f'libmc3hip_{name}.so')
    subprocess.check_call([b.hipcc()] + b.LDFLAGS + ['-o', lib] + objs)
    print(lib)


if __name__ == '__main__':
    main()
