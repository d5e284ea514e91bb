"""The test suite's side of the library's launch ledger (include/mc3hip.h
mc3hip_debug_launch_counts): every process of a test session -- pytest itself and the rank /
worker scripts it starts -- leaves its ledger in $MC3HIP_LEDGER_DIR when it ends; the sum, set
against the kernel instances libmc3hip.so carries (their handles, as `nm` lists them),
is the instance-coverage report of tests/test_zz_instance_coverage.py."""
import atexit
import os
import re
import subprocess

ENV = 'MC3HIP_LEDGER_DIR'
_installed = False


def dump():
    """This process's ledger -> $MC3HIP_LEDGER_DIR/<pid>.txt (nothing if the library was never
    loaded or launched nothing)."""
    where = os.environ.get(ENV)
    if not where:
        return
    try:
        from mc3_amd import _lib
        if _lib._LIB is None:
            return
        rows = _lib.launch_counts()
        if rows:
            os.makedirs(where, exist_ok=True)
            tmp = os.path.join(where, f'.{os.getpid()}.tmp')
            with open(tmp, 'w') as f:
                for n, plain, off, name, switches in rows:
                    f.write(' '.join([str(n), str(plain), hex(off), name] + switches) + '\n')
            os.replace(tmp, os.path.join(where, f'{os.getpid()}.txt'))
    except Exception as exc:            # (a ledger must never fail a run)
        print(f'ledger: {exc!r}')


def install():
    global _installed
    if not _installed and os.environ.get(ENV):
        atexit.register(dump)
        _installed = True


def collected():
    """{stub offset or 'jit:<kernel>': [launches, launches without a switch, name, {switches}]}
    summed over this process and every file of $MC3HIP_LEDGER_DIR."""
    total = {}

    def add(n, plain, off, name, switches):
        key = name if off == 0 else off
        row = total.setdefault(key, [0, 0, name, set()])
        row[0] += n
        row[1] += plain
        row[3].update(switches)

    from mc3_amd import _lib
    if _lib._LIB is not None:
        for row in _lib.launch_counts():
            add(*row)
    where = os.environ.get(ENV)
    if where and os.path.isdir(where):
        for fn in sorted(os.listdir(where)):
            if fn.endswith('.txt') and fn != f'{os.getpid()}.txt':
                with open(os.path.join(where, fn)) as f:
                    for line in f:
                        n, plain, off, name, *sw = line.split()
                        add(int(n), int(plain), int(off, 16), name, sw)
    return total


def instances(lib_path=None):
    """{handle offset: demangled kernel instance} of every kernel libmc3hip.so carries.  What the
    host passes to a launch is the kernel's HANDLE -- a data object of the kernel's own name --
    and the ledger records its offset; the host stub of the same instance
    (`__device_stub__<name>`) is a function beside it.  An instance = a name that has both."""
    from mc3_amd import _lib
    lib_path = lib_path or _lib.LIB_PATH
    out = subprocess.run(['nm', '-C', '--defined-only', lib_path], capture_output=True,
                         text=True, check=True).stdout
    stubs, handles = set(), {}
    for line in out.splitlines():
        parts = line.split(' ', 2)
        if len(parts) != 3:
            continue
        addr, kind, name = parts
        if '__device_stub__' in name:
            stubs.add(name.replace('mc3hip::__device_stub__', 'mc3hip::')
                      .replace('__device_stub__', ''))
        elif kind in 'VvDdBbRr':
            handles[name] = int(addr, 16)
    return {handles[name]: name for name in stubs if name in handles}


def family(name):
    """'mc3hip::lean_demc_kernel<3, false>(...)' -> 'lean_demc_kernel'."""
    m = re.search(r'(\w+)(?:<|\()', name.replace('void ', ''))
    return m.group(1) if m else name


def report(total=None):
    """(text, never) -- the coverage table and the instances never launched."""
    total = collected() if total is None else total
    inst = instances()
    lines, never, steered = [], [], []
    fams = {}
    for off, name in sorted(inst.items(), key=lambda kv: kv[1]):
        row = total.get(off)
        fam = fams.setdefault(family(name), [0, 0, 0])
        fam[0] += 1
        if row is None:
            never.append(name)
        else:
            fam[1] += 1
            if row[1] == 0:
                steered.append((name, sorted(row[3])))
            else:
                fam[2] += 1
    lines.append(f'kernel instances in libmc3hip.so: {len(inst)}; launched in this session: '
                 f'{len(inst) - len(never)}; of those by the dispatch left alone (no MC3HIP_* '
                 f'switch set): {len(inst) - len(never) - len(steered)}; never: {len(never)}')
    lines.append('')
    lines.append(f'{"family":34s} {"instances":>9s} {"launched":>9s} {"unsteered":>9s}')
    for name, (n, hit, plain) in sorted(fams.items()):
        lines.append(f'{name:34s} {n:9d} {hit:9d} {plain:9d}')
    jit = sorted(k for k in total if isinstance(k, str))
    lines.append('')
    lines.append(f'loops specialised at run time (not part of the library): {len(jit)}')
    for k in jit:
        lines.append(f'  {total[k][0]:6d}  {k}')
    lines.append('')
    lines.append(f'launched only with a switch set: {len(steered)}')
    for name, sw in steered:
        lines.append(f'  {name}    [{" ".join(sw)}]')
    lines.append('')
    lines.append(f'never launched: {len(never)}')
    for name in never:
        lines.append(f'  {name}')
    lines.append('')
    stray = sorted((k, v) for k, v in total.items() if not isinstance(k, str) and k not in inst)
    if stray:
        lines.append(f'ledger rows that match no instance of the library: {len(stray)}')
        for k, v in stray:
            lines.append(f'  {v[0]:8d}  {hex(k)}  {v[2]}')
        lines.append('')
    lines.append('launches per instance:')
    for off, name in sorted(inst.items(), key=lambda kv: kv[1]):
        if off in total:
            lines.append(f'  {total[off][0]:8d} {total[off][1]:8d}  {name}')
    return '\n'.join(lines) + '\n', never
