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


STAMP = LIB + '.srchash'


def source_hash():
    """Content hash of everything the library is compiled from (and of the flags)."""
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for d in DEPS:
        with open(d, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def is_stale():
    """The library is rebuilt when it is missing or was built from other sources: decided
    by content, not by modification times (a copy of the tree to another machine need not
    preserve those, and a spurious rebuild costs minutes)."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [hipcc()] + FLAGS + ['-o', LIB, SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, 'w') as f:
        f.write(source_hash() + '\n')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
    print(LIB)
