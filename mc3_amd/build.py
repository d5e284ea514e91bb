"""Build libmc3hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m mc3_amd.build [--force]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'mc3hip.hip')
DEPS = [
    SRC,
    os.path.join(HERE, 'csrc', 'chain_kernels.h'),
    os.path.join(HERE, 'csrc', 'philox.h'),
    os.path.join(HERE, '..', 'include', 'mc3hip.h'),
    os.path.join(HERE, '..', 'include', 'mc3hip_detmath.h'),
]
LIB = os.path.join(HERE, 'libmc3hip.so')

# -ffp-contract=off is part of the arithmetic contract (include/mc3hip_detmath.h):
# fused multiply-adds happen only where the source spells them out.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
         '-fPIC', '-shared']


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: cannot build libmc3hip.so')
    return exe


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [hipcc()] + FLAGS + ['-o', LIB, SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
    print(LIB)
