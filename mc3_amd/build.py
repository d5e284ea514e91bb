"""Build libmc3hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m mc3_amd.build [--force] [--resource-usage] [--dev]

--dev: a DEVELOPMENT build beside the product, mc3_amd/libmc3hip_dev.so (-DMC3HIP_DEV: the A/B
switches of experiments, csrc/ctx.h dev_env_text, and the diagnostic macros of csrc/core.h exist
only there); MC3HIP_LIBRARY=<path> makes the package load it.

The library is several translation units (csrc/mc3hip.hip + one csrc/k_*.hip per kernel
family) compiled in parallel and linked into ONE shared object.  An object is recompiled
only when the content of its source, of a header it includes (transitively) or the flags
changed, so touching one kernel family rebuilds one unit, not the library.
"""
import concurrent.futures
import fcntl
import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
# (the slowest first: the pool starts them in this order)
UNITS = ['k_lean.hip', 'k_lean6.hip', 'k_lean7.hip', 'k_lean2.hip', 'k_general_w1.hip', 'k_wavelet_w1.hip', 'k_wavelet_w1b.hip',
         'k_general_w4.hip', 'k_wavelet_large.hip', 'k_resident.hip', 'k_lean5.hip',
         'k_wavelet_w8cg.hip', 'k_wavelet_w4.hip', 'k_wavelet_w8mc.hip', 'k_wavelet_w8pp.hip', 'k_wavelet_w8cw.hip',
         'k_lean4.hip', 'k_wavelet_w8.hip', 'k_lean3.hip', 'k_tiled.hip', 'mc3hip.hip']
SRC = os.path.join(CSRC, 'mc3hip.hip')
LIB = os.path.join(HERE, 'libmc3hip.so')
STAMP = LIB + '.srchash'
OBJDIR = os.path.join(HERE, '..', 'build', 'obj')

# -ffp-contract=off is part of the arithmetic contract (include/mc3hip_detmath.h):
# fused multiply-adds happen only where the source spells them out.
# --offload-compress: the gfx950 code objects are stored compressed in the library (the HIP
# runtime unpacks them when it loads the library): 15.6 -> ~6 MB for the 628 kernel instances.
CFLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
          '-fvisibility=hidden', '--offload-compress']
LDFLAGS = ['--offload-arch=gfx950', '-shared', '-fPIC', '--offload-compress']
FLAGS = CFLAGS          # (what tools that build a diagnostic variant start from)

_INCLUDE = re.compile(r'^\s*#\s*include\s*"([^"]+)"', re.M)


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: cannot build libmc3hip.so')
    return exe


def dependencies(path, seen=None):
    """`path` and every file it includes with "...", transitively (sorted, absolute)."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen:
        return seen
    seen.add(path)
    with open(path, errors='ignore') as f:
        text = f.read()
    for inc in _INCLUDE.findall(text):
        dep = os.path.normpath(os.path.join(os.path.dirname(path), inc))
        if os.path.exists(dep):
            dependencies(dep, seen)
    return seen


def unit_hash(unit, extra=()):
    """Content hash of everything one object is compiled from (and of the flags)."""
    h = hashlib.sha256(' '.join(list(CFLAGS) + list(extra)).encode())
    for d in sorted(dependencies(os.path.join(CSRC, unit))):
        h.update(os.path.relpath(d, HERE).encode())
        with open(d, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def source_hash(extra=()):
    """Content hash of the whole library: its units' hashes and the link flags."""
    h = hashlib.sha256(' '.join(LDFLAGS).encode())
    for u in UNITS:
        h.update(unit_hash(u, extra).encode())
    return h.hexdigest()


def is_stale():
    """The library is rebuilt when it is missing or was built from other sources: decided
    by content, not by modification times (a copy of the tree to another machine need not
    preserve those, and a spurious rebuild costs minutes)."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def _atomic_write(path, text):
    tmp = f'{path}.tmp{os.getpid()}'
    with open(tmp, 'w') as f:
        f.write(text)
    os.replace(tmp, path)


def _compile(unit, objdir, extra, verbose):
    """One object, recompiled only when its hash changed.  The hash is taken BEFORE the
    compiler runs, so a source edited meanwhile is seen as newer next time."""
    want = unit_hash(unit, extra)
    obj = os.path.join(objdir, os.path.splitext(unit)[0] + '.o')
    stamp = obj + '.srchash'
    if os.path.exists(obj) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == want:
                return obj, False
    tmp = f'{obj}.tmp{os.getpid()}'
    cmd = [hipcc()] + CFLAGS + list(extra) + ['-c', os.path.join(CSRC, unit), '-o', tmp]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, obj)
    _atomic_write(stamp, want + '\n')
    return obj, True


def build(force=False, verbose=False, lib=LIB, extra=(), jobs=None):
    """Compile the stale units in parallel and link.  `extra`: more compiler flags (a
    diagnostic variant of the library under another name keeps its own objects)."""
    extra = tuple(extra)
    product = os.path.abspath(lib) == os.path.abspath(LIB)
    if product and not force and not is_stale():
        return lib
    objdir = os.path.normpath(OBJDIR if product else
                              OBJDIR + '_' + hashlib.sha256(' '.join(extra).encode())
                              .hexdigest()[:10])
    os.makedirs(objdir, exist_ok=True)
    # one builder at a time (two processes importing a stale tree at once)
    with open(os.path.join(objdir, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if product and not force and not is_stale():
            return lib
        want = source_hash(extra)
        if force:
            for f in os.listdir(objdir):
                if f.endswith('.srchash'):
                    os.remove(os.path.join(objdir, f))
        jobs = jobs or min(len(UNITS), os.cpu_count() or 1)
        with concurrent.futures.ThreadPoolExecutor(jobs) as pool:
            objs = [o for o, _ in pool.map(lambda u: _compile(u, objdir, extra, verbose), UNITS)]
        tmp = f'{lib}.tmp{os.getpid()}'
        cmd = [hipcc()] + LDFLAGS + ['-o', tmp] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(tmp, lib)                      # the library first, then its stamp
        if product:
            _atomic_write(STAMP, want + '\n')
    return lib


PROBES = os.path.normpath(os.path.join(HERE, '..', 'tools', 'probes'))
# measurement probes bench.py and the GPU tests run as child processes (built with the library,
# they travel to the GPU box as files): the streaming-read probe behind roofline.peak_measured,
# and rocRAND's own Philox state for the identity test of csrc/philox.h
PROBE_UNITS = {'hbm_read_probe': ['-O3'], 'rocrand_philox_probe': ['-O2'],
               'sync_stall_probe': ['-O2'], 'fp64_operand_probe': ['-O2'],
               'wave_sum_probe': ['-O2']}


def probe_path(name):
    return os.path.join(PROBES, '_bin', name)


def build_probes(verbose=False):
    """tools/probes/<name>.hip -> tools/probes/_bin/<name>, recompiled when the source changed."""
    os.makedirs(os.path.join(PROBES, '_bin'), exist_ok=True)
    for name, flags in PROBE_UNITS.items():
        src, exe = os.path.join(PROBES, name + '.hip'), probe_path(name)
        with open(src, 'rb') as f:
            want = hashlib.sha256(f.read() + ' '.join(flags).encode()).hexdigest()
        stamp = exe + '.srchash'
        if os.path.exists(exe) and os.path.exists(stamp):
            with open(stamp) as f:
                if f.read().strip() == want:
                    continue
        cmd = [hipcc(), '--offload-arch=gfx950'] + flags + ['-o', exe + '.tmp', src]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(exe + '.tmp', exe)
        _atomic_write(stamp, want + '\n')


def resource_usage(out=sys.stdout):
    """Registers, spills, scratch and LDS of every kernel instance, from the compiler's
    remarks (recompiles every unit; nothing is linked)."""
    for unit in UNITS:
        cmd = [hipcc()] + CFLAGS + ['-Rpass-analysis=kernel-resource-usage', '-c',
                                   os.path.join(CSRC, unit), '-o', os.devnull]
        res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True, check=True)
        out.write(f'## {unit}\n{res.stderr}')


if __name__ == '__main__':
    if '--resource-usage' in sys.argv:
        resource_usage()
    elif '--dev' in sys.argv:
        build(force='--force' in sys.argv, verbose=True,
              lib=os.path.join(HERE, 'libmc3hip_dev.so'), extra=['-DMC3HIP_DEV'])
    else:
        build(force='--force' in sys.argv, verbose=True)
        build_probes(verbose=True)
        print(LIB)
