"""The drop-in boundary without a GPU: libmc3hip.so loads, exports every entry
point include/mc3hip.h declares, the ctypes mirror of its structs has the C
layout, and the product never reaches into oracle/.  No compute calls here."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'mc3hip.h')


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mc3hip_\w+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from mc3_amd import _lib
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f'{name} declared in include/mc3hip.h but not exported'
    assert sorted(_lib.SYMBOLS) == names, 'mc3_amd/_lib.py binds a different set of symbols'


def test_library_exports_no_other_functions():
    """Built with -fvisibility=hidden: the only functions the dynamic symbol table
    offers are the C-ABI's (kernel handles are data objects, libstdc++ templates weak)."""
    from mc3_amd import _lib
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    functions = [line.split()[2] for line in out.splitlines()
                 if len(line.split()) == 3 and line.split()[1] == 'T']
    assert sorted(functions) == declared_functions()


def test_error_reporting_without_a_gpu():
    """Argument errors are reported through return codes + mc3hip_last_error, never
    by aborting; with no GPU present creation fails with a HIP error code."""
    from mc3_amd import _lib
    lib = _lib.load()
    assert lib.mc3hip_create(None, None) == _lib.ERR_ARG
    assert b'null' in lib.mc3hip_last_error()
    P = _lib.Problem()
    P.struct_bytes = 1
    ctx = ctypes.c_void_p()
    assert lib.mc3hip_create(ctypes.byref(P), ctypes.byref(ctx)) == _lib.ERR_ARG
    assert b'size mismatch' in lib.mc3hip_last_error()
    assert lib.mc3hip_run(None, 1) == _lib.ERR_ARG
    lib.mc3hip_destroy(None)


def test_ctypes_structs_match_the_c_layout(tmp_path):
    from mc3_amd import _lib
    src = tmp_path / 'layout.c'
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "mc3hip.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(mc3hip_problem),'
        ' offsetof(mc3hip_problem, M0), offsetof(mc3hip_problem, gamma),'
        ' offsetof(mc3hip_problem, uncert), sizeof(mc3hip_plan),'
        ' offsetof(mc3hip_plan, persistent)); return 0;}\n')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(_lib.Problem), _lib.Problem.M0.offset, _lib.Problem.gamma.offset,
            _lib.Problem.uncert.offset, ctypes.sizeof(_lib.Plan), _lib.Plan.persistent.offset]
    assert got == want


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use
    oracle/: nothing under mc3_amd/ (Python or C++) mentions it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'mc3_amd')):
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.cpp')):
                text = open(os.path.join(dirpath, f), errors='ignore').read()
                if re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M) or \
                        'liboracle' in text or 'oracle/' in text.replace('oracle/oracle_lockstep.c follows', ''):
                    bad.append(os.path.join(dirpath, f))
    assert bad == []


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mc3_amd import _lib
    monkeypatch.setattr(_lib, '_LIB', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(ImportError, match='no CPU fallback'):
        _lib.load()


def test_jit_compiles_a_request_without_a_gpu(tmp_path, monkeypatch):
    """mc3_amd/jit.py: hipcc cross-compiles the generated unit for gfx950 here too; the code
    object is cached under a key of the request, the headers and the flags."""
    from mc3_amd import jit
    monkeypatch.setenv('MC3HIP_JIT_CACHE', str(tmp_path))
    req = 'lean_demc_body<206, false, false, false, false, false, false>'
    image = jit.code_object(req)
    # (a code-object bundle, or the bare ELF: hipModuleLoadData takes either)
    assert image[:4] in (b'\x7fELF', b'__CL') and len(image) > 4000
    files = os.listdir(str(tmp_path))
    assert len(files) == 1
    assert jit.code_object(req) == image                      # (from the file)
    assert jit._key(req) != jit._key(req.replace('206', '207'))


def test_a_specialised_loop_that_fails_the_audit_is_not_installed(tmp_path, monkeypatch, capsys):
    """mc3_amd/jit.py audits the hidden loads of what it compiled (isa_audit.py); a loop in which
    the allocator touched an in-flight register is neither cached nor installed: the context keeps
    the general chain, with a word on stderr."""
    from mc3_amd import isa_audit, jit
    monkeypatch.setenv('MC3HIP_JIT_CACHE', str(tmp_path))
    monkeypatch.setattr(isa_audit, 'violations',
                        lambda text: ([('mc3_jit_kernel', '.LBB0_1', 'v_mov_b32 v1, v10', [10])], 3))

    class Engine:
        installed = None

        def jit_request(self):
            return 'lean_snooker_body<205, false, false, false, false, false, false>'

        def jit_install(self, image, name):
            self.installed = image
    eng = Engine()
    assert jit.specialise(eng) == '' and eng.installed is None
    assert os.listdir(str(tmp_path)) == []
    assert 'hidden load' in capsys.readouterr().err


def test_jit_cache_directory_must_be_private(tmp_path, monkeypatch):
    """a cached code object is loaded into the process without an audit, so the
    cache is accepted only if it is a directory of this user that group / others cannot write
    to; a planted or open directory is passed over for a private one."""
    from mc3_amd import jit
    good = tmp_path / 'cache'
    monkeypatch.setenv('MC3HIP_JIT_CACHE', str(good))
    assert jit.cache_dir() == str(good) and os.stat(good).st_mode & 0o077 == 0
    os.chmod(good, 0o777)                           # (anybody could plant <key>.hsaco there)
    monkeypatch.setattr(jit.tempfile, 'gettempdir', lambda: str(tmp_path))
    d = jit.cache_dir()
    assert d != str(good) and os.stat(d).st_uid == os.getuid() and os.stat(d).st_mode & 0o022 == 0
    # the fallback under the temporary directory pre-created open by somebody: passed over too
    os.chmod(d, 0o777)
    monkeypatch.setattr(jit, '_FALLBACK', None)
    e = jit.cache_dir()
    assert e not in (str(good), d) and os.stat(e).st_mode & 0o077 == 0
    # the compiler's version is part of the key
    monkeypatch.setattr(jit, '_HIPCC_VERSION', 'hipcc 1')
    k1 = jit._key('lean_demc_body<6, false, false, false, false, false, false>')
    monkeypatch.setattr(jit, '_HIPCC_VERSION', 'hipcc 2')
    assert jit._key('lean_demc_body<6, false, false, false, false, false, false>') != k1


def test_launch_ledger_without_a_gpu():
    """The launch ledger (mc3hip_debug_launch_counts): empty before any launch, argument errors
    by code; and tests/ledger.py finds every kernel instance of the library by its handle --
    what the GPU suite's last test sets the ledgers of its processes against."""
    from mc3_amd import _lib
    lib = _lib.load()
    assert _lib.launch_counts() == []
    assert lib.mc3hip_debug_launch_counts(None, 0, None) == _lib.ERR_ARG
    assert lib.mc3hip_debug_last_kernel(None, None, 0) == _lib.ERR_ARG
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import ledger
    inst = ledger.instances()
    fams = {ledger.family(n) for n in inst.values()}
    assert len(inst) >= 500 and {'lean_demc_kernel', 'resident_async_kernel',
                                 'wavelet_pipe_kernel', 'persist_kernel'} <= fams
    text, never = ledger.report({})
    assert len(never) == len(inst) and 'never launched' in text
