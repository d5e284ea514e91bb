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
    flags = [x for x in b.CFLAGS if x not in ('-fPIC', '-fvisibility=hidden')]
    subprocess.run([b.hipcc()] + flags + ['--save-temps', '-I', CSRC, '-c', 'jit.hip', '-o',
                                          '/dev/null'], cwd=d, check=True, capture_output=True)
    (path,) = glob.glob(os.path.join(d, '*gfx950.s'))
    with open(path) as f:
        return f.read()


@pytest.fixture(scope='module')
def isa(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp('isa'))
    with concurrent.futures.ThreadPoolExecutor(8) as pool:
        units = pool.map(lambda u: _isa(u, tmp), UNITS)
        jits = pool.map(lambda kr: _isa_request(kr[0], kr[1], tmp), enumerate(JIT_REQUESTS))
        out = dict(zip(UNITS, units))
        out.update({f'jit: {r}': t for r, t in zip(JIT_REQUESTS, jits)})
        return out


def _kernels(text):
    """(name, lines of the body) of every kernel function in an assembly file."""
    lines = text.split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^(_Z\w+|mc3_jit_kernel):', l)]
    for k, i in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(lines)
        body = lines[i:end]
        stop = next((j for j, l in enumerate(body) if l.startswith('.Lfunc_end')), len(body))
        yield lines[i].split(':')[0], body[:stop]


def _regs(operand_text, bank):
    """Register numbers of one bank ('v' / 'a') named in an instruction's operands."""
    out = set()
    for m in re.finditer(r'\b%s(\d+)\b|\b%s\[(\d+):(\d+)\]' % (bank, bank), operand_text):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def test_no_scratch_in_the_hand_laid_out_loops(isa):
    for unit, text in isa.items():
        if unit.startswith('jit: '):          # (a ten-column tile on one wave may spill: slow, not wrong)
            continue
        meta = text[text.index('amdhsa.kernels:'):]
        for blk in meta.split('  - .agpr_count')[1:]:
            name = re.search(r'\.name:\s+(\S+)', blk).group(1)
            scratch = int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk).group(1))
            assert scratch == 0, f'{unit}: {name} carries {scratch} B of scratch'


def _blocks(body):
    """Basic blocks of a kernel: {label: (instructions, successors)}; an instruction is
    (in_inline_asm, mnemonic, operand text)."""
    blocks, order = {}, []
    label, cur, inasm = 'entry', [], False

    def close(nxt):
        blocks[label] = [cur, nxt]
        order.append(label)
    for line in body[1:]:
        if '#ASMSTART' in line:
            inasm = True
            continue
        if '#ASMEND' in line:
            inasm = False
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', line)
        if m:
            close(m.group(1))
            label, cur = m.group(1), []
            continue
        code = line.split(';')[0].strip()
        if not code or code.endswith(':') or code.startswith('.'):
            continue
        op, _, rest = code.partition(' ')
        cur.append((inasm, op, rest))
    close(None)
    succ = {}
    for k, lab in enumerate(order):
        ins, fall = blocks[lab]
        out = set()
        ends = False
        for _, op, rest in ins:
            if op == 's_branch':
                out.add(rest.strip())
                ends = True
            elif op.startswith('s_cbranch'):
                out.add(rest.strip())
            elif op == 's_endpgm':
                ends = True
        if not ends and fall is not None:
            out.add(fall)
        succ[lab] = out
    return {lab: blocks[lab][0] for lab in order}, succ, order


def test_async_destinations_are_left_alone_until_their_wait(isa):
    """Forward data flow over the kernel's control-flow graph: the set of VGPRs an inline-assembly
    load is still on its way to; any other instruction that names one of them on ANY path
    before an inline-assembly wait is an error."""
    checked = 0
    for unit, text in isa.items():
        for name, body in _kernels(text):
            blocks, succ, order = _blocks(body)
            state = {lab: None for lab in order}        # pending set at block entry
            state[order[0]] = frozenset()
            work = [order[0]]
            while work:
                lab = work.pop()
                pending = set(state[lab])
                for inasm, op, rest in blocks[lab]:
                    if inasm:
                        if op == 'global_load_dwordx2' and not rest.lstrip().startswith('a['):
                            pending |= _regs(rest.split(',')[0], 'v')
                            checked += 1
                        elif op == 's_waitcnt':
                            pending = set()
                        continue
                    if op == 's_waitcnt' and 'vmcnt(0)' in rest:
                        pending = set()       # (a full wait the compiler placed: everything is in)
                        continue
                    # what the instruction READS: every operand of a store / compare / atomic,
                    # the operands behind the destination otherwise.  (Writes are not flagged:
                    # the structurised control flow has statically possible, dynamically dead
                    # paths -- a "break" flag routed through the loop latch -- on which the
                    # allocator rightly reuses a register whose value is dead there.)
                    reads_all = op.startswith(('global_store', 'scratch_store', 'buffer_store',
                                               'flat_store', 'ds_write', 'ds_store', 'v_cmp',
                                               'global_atomic', 'flat_atomic'))
                    srcs = rest if reads_all else rest.partition(',')[2]
                    touched = _regs(srcs, 'v') & pending
                    assert not touched, (
                        f'{unit}: {name}: {lab}: `{op} {rest}` reads v{sorted(touched)} while an '
                        'inline-assembly load into it is still in flight')
                    if not reads_all:         # (overwritten: the allocator holds the value dead)
                        pending -= _regs(rest.partition(',')[0], 'v')
                for nxt in succ[lab]:
                    if nxt not in state:
                        continue
                    new = frozenset(pending) if state[nxt] is None else state[nxt] | pending
                    if new != state[nxt]:
                        state[nxt] = new
                        work.append(nxt)
    assert checked > 50           # (the scan found the loads it is about)


def test_the_scan_sees_the_specialised_kernels(isa):
    for unit, text in isa.items():
        if unit.startswith('jit: '):
            names = [name for name, _ in _kernels(text)]
            assert names == ['mc3_jit_kernel'], (unit, names)
            assert 'global_load_dwordx2' in text and '#ASMSTART' in text, unit
