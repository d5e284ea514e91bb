"""Build-time guards for the loads the compiler does not see (mc3_amd/csrc/handoff.h).

The hand-laid-out loops ask for Z rows an iteration (or two) ahead with loads issued from
inline assembly and retire them with hand-counted ``s_waitcnt vmcnt(N)``.  That is correct only
while the register allocator leaves the destinations alone between the two -- no copy, no
spill, no accumulation-register move -- which nothing in the language promises: a compiler bump
or a change of register pressure could break it silently (it did once, found by a fuzz test:
handoff.h).  These tests recompile the units that hold such loops with ``--save-temps`` (hipcc
cross-compiles without a GPU) and check the ISA:

* ``load_published_async`` (ordinary registers): between the load and the next inline-assembly
  wait, on any path of the control-flow graph, no instruction reads a destination register
  (a copy, a spill, an accumulation-register move or a use);
* none of these kernels carries scratch;
* the same scan over run-time specialised instances (the source mc3_amd/jit.py would generate for
  design matrices and long polynomials, compiled here as it would be on the user's machine).
"""
import concurrent.futures
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'mc3_amd', 'csrc')
UNITS = ('k_lean2.hip', 'k_lean3.hip', 'k_lean4.hip', 'k_lean6.hip', 'k_lean7.hip')
# (no hidden loads, but a scratch budget of its own: below)
PIPE_UNIT = 'k_wavelet_w8pp.hip'
pytestmark = pytest.mark.timeout(900)


def _isa(unit, tmp):
    from mc3_amd import build as b
    d = os.path.join(tmp, unit[:-4])
    os.makedirs(d)
    subprocess.run([b.hipcc()] + b.CFLAGS + ['--save-temps', '-c', os.path.join(CSRC, unit),
                                             '-o', '/dev/null'], cwd=d, check=True,
                   capture_output=True)
    (path,) = glob.glob(os.path.join(d, '*gfx950.s'))
    with open(path) as f:
        return f.read()


# ... and run-time specialised instances of the same loops (mc3_amd/jit.py compiles them on the
# user's machine: what the library would ask for for these shapes, compiled here the same way) --
# design matrices whose tile fills the registers are where an allocator is most tempted
JIT_REQUESTS = (
    'lean_snooker_body<210, false, false, false, false, false, false, 4> /*threads=256*/',
    'lean_snooker_body<208, true, false, false, false, false, false, 4> /*threads=256*/',
    'lean_snooker_body<8, false, false, false, false, false, false>',
    'lean_snooker_body<204, false, true, false, false, false, false>',
    'lean_demc_lag_body<210, false, false, false>',
    'lean_demc_lag_body<205, false, true, false>',
    'lean_demc_lag_body<7, false, false, false>',
)


def _isa_request(k, request, tmp):
    from mc3_amd import build as b, jit
    d = os.path.join(tmp, f'jit{k}')
    os.makedirs(d)
    m = re.search(r'/\*threads=(\d+)\*/', request)
    with open(os.path.join(d, 'jit.hip'), 'w') as f:
        f.write(jit._SOURCE % (int(m.group(1)) if m else 64, jit.KERNEL_NAME, request))
    flags = [x for x in b.CFLAGS if x not in ('-fPIC', '-fvisibility=hidden', '--offload-compress')]
    subprocess.run([b.hipcc()] + flags + ['--save-temps', '-I', CSRC, '-c', 'jit.hip', '-o',
                                          '/dev/null'], cwd=d, check=True, capture_output=True)
    (path,) = glob.glob(os.path.join(d, '*gfx950.s'))
    with open(path) as f:
        return f.read()


@pytest.fixture(scope='module')
def isa(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp('isa'))
    with concurrent.futures.ThreadPoolExecutor(8) as pool:
        units = pool.map(lambda u: _isa(u, tmp), UNITS + (PIPE_UNIT,))
        jits = pool.map(lambda kr: _isa_request(kr[0], kr[1], tmp), enumerate(JIT_REQUESTS))
        out = dict(zip(UNITS + (PIPE_UNIT,), units))
        out.update({f'jit: {r}': t for r, t in zip(JIT_REQUESTS, jits)})
        return out


from mc3_amd import isa_audit


def test_no_scratch_in_the_hand_laid_out_loops(isa):
    for unit, text in isa.items():
        if unit.startswith('jit: '):          # (a ten-column tile on one wave may spill: slow, not wrong)
            continue
        for name, scratch in isa_audit.scratch_bytes(text).items():
            if unit == PIPE_UNIT and 'ILi0E' in name:
                # (the pipelined wavelet kernel -- eleven waves, 168 registers each: C4's kernel --
                # with the GENERIC polynomial, whose coefficient loop costs the compute waves four
                # registers more than there are: 20 B, as in the nine-wave kernels; 28 B since round
                # 6, where the give-up decision of a launch became one per workgroup)
                assert scratch <= 32, f'{unit}: {name} carries {scratch} B of scratch'
                continue
            if 'wide_kernel' in name and 'ELi8ELi16E' in name:
                # (sixteen 1024-point tiles in the registers of eight waves, kernels_lean.h
                # lean_*_wide_kernel<MT, 8, 16>: 192 registers of data beside the chain's own
                # state -- a few dozen values spill; 1.7 / 2.8 us per iteration at 16 384 points
                # against the 6.7 / 7.9 of the plans these shapes ran before)
                assert scratch <= 320, f'{unit}: {name} carries {scratch} B of scratch'
                continue
            assert scratch == 0, f'{unit}: {name} carries {scratch} B of scratch'


def test_async_destinations_are_left_alone_until_their_wait(isa):
    """Forward data flow over the kernel's control-flow graph (mc3_amd/isa_audit.py, the check
    mc3_amd/jit.py applies to what it compiles at run time): the set of VGPRs an inline-assembly
    load is still on its way to; any other instruction that names one of them on ANY path
    before an inline-assembly wait is an error."""
    checked = 0
    for unit, text in isa.items():
        found, n = isa_audit.violations(text)
        checked += n
        assert not found, f'{unit}: ' + '; '.join(
            f'{k}: {lab}: `{ins}` reads v{regs} while an inline-assembly load into it is still '
            'in flight' for k, lab, ins, regs in found[:5])
    assert checked > 50           # (the scan found the loads it is about)


def test_the_scan_sees_the_specialised_kernels(isa):
    for unit, text in isa.items():
        if unit.startswith('jit: '):
            names = [name for name, _ in isa_audit._kernels(text)]
            assert names == ['mc3_jit_kernel'], (unit, names)
            assert 'global_load_dwordx2' in text and '#ASMSTART' in text, unit


def test_the_scan_catches_a_copy_of_an_in_flight_register():
    """The check itself: a listing in which the allocator copies a hidden load's destination before
    the wait."""
    text = """_Z4testv:
\t#ASMSTART
\tglobal_load_dwordx2 v[10:11], v[2:3], off sc1
\t#ASMEND
\tv_mov_b32_e32 v20, v10
\t#ASMSTART
\ts_waitcnt vmcnt(1)
\t#ASMEND
\ts_endpgm
.Lfunc_end0:
"""
    found, n = isa_audit.violations(text)
    assert n == 1 and len(found) == 1 and found[0][3] == [10]
    ok = text.replace('v_mov_b32_e32 v20, v10', 'v_mov_b32_e32 v20, v12')
    assert isa_audit.violations(ok)[0] == []


def test_the_scan_counts_partial_waits_in_issue_order():
    """``s_waitcnt vmcnt(N)`` says "at most N still in flight", in issue order, loads
    and stores alike -- it retires a hidden load only if N operations were issued AFTER it.
    Synthetic listings: (a) load, store, vmcnt(1): the load is in (the store may be out); (b)
    load, then a YOUNGER hidden load, vmcnt(1): the older one is in, the younger one is not --
    a read of its destination is flagged; (c) two paths that join, one of which issued
    nothing after the load: the join keeps it pending through vmcnt(1); (d) the compiler's
    own partial wait counts as well, a wait on another counter does not."""
    def listing(*body):
        return '_Z4testv:\n' + ''.join(f'{ln}\n' if ln.endswith(':') else f'\t{ln}\n'
                                        for ln in body) + '\ts_endpgm\n.Lfunc_end0:\n'

    asm = lambda ins: ('#ASMSTART', ins, '#ASMEND')             # noqa: E731
    load_a = asm('global_load_dwordx2 v[10:11], v[2:3], off sc1')
    load_b = asm('global_load_dwordx2 v[12:13], v[2:3], off sc1')
    store = asm('global_store_dwordx2 v[4:5], v[6:7], off sc1')
    wait1 = asm('s_waitcnt vmcnt(1)')
    use_a, use_b = 'v_add_f64 v[20:21], v[10:11], v[10:11]', 'v_add_f64 v[22:23], v[12:13], v[12:13]'
    # (a)
    assert isa_audit.violations(listing(*load_a, *store, *wait1, use_a))[0] == []
    # (b)
    found, n = isa_audit.violations(listing(*load_a, *load_b, *wait1, use_a, use_b))
    assert n == 2 and [f[3] for f in found] == [[12, 13]]
    assert isa_audit.violations(listing(*load_a, *load_b, *asm('s_waitcnt vmcnt(0)'),
                                        use_a, use_b))[0] == []
    # (the compiler's own store after the load counts too)
    assert isa_audit.violations(listing(*load_a, 'global_store_dwordx2 v[4:5], v[6:7], off',
                                        *wait1, use_a))[0] == []
    # (c) the store only on one of two paths into the wait
    joined = listing(*load_a, 's_cbranch_scc0 .LBB0_2', *store, '.LBB0_2:', *wait1, use_a)
    assert [f[3] for f in isa_audit.violations(joined)[0]] == [[10, 11]]
    # (d)
    assert isa_audit.violations(listing(*load_a, *store, 's_waitcnt vmcnt(1)', use_a))[0] == []
    assert isa_audit.violations(listing(*load_a, *store, 's_waitcnt vmcnt(2)', use_a))[0] != []
    assert isa_audit.violations(listing(*load_a, *store, *asm('s_waitcnt lgkmcnt(0)'),
                                        use_a))[0] != []
    assert isa_audit.violations(listing(*load_a, 's_waitcnt 0x0f70', use_a))[0] == []
    assert isa_audit.violations(listing(*load_a, 's_waitcnt 0xc07f', use_a))[0] != []
    # (e) the way out of a loop routed through its latch (a backward jump taken with the loads
    # still pending) is not followed into the next iteration; the same read on the way FORWARD is
    loop = listing('.LBB0_1:', 'v_mov_b32_e32 v30, v10', *load_a, 's_cbranch_scc1 .LBB0_1',
                   *store, *wait1, use_a, 's_branch .LBB0_1')
    assert isa_audit.violations(loop)[0] == []
    fwd = listing('.LBB0_1:', *load_a, 's_cbranch_scc1 .LBB0_3', *store, *wait1, use_a,
                  's_branch .LBB0_1', '.LBB0_3:', 'v_mov_b32_e32 v30, v10')
    assert [f[3] for f in isa_audit.violations(fwd)[0]] == [[10]]
