"""Static check of the loads the compiler does not see (mc3_amd/csrc/handoff.h
load_published_async / published_arrived): the hand-laid-out loops ask for Z rows an iteration
ahead with loads issued from inline assembly and retire them with hand-counted
``s_waitcnt vmcnt(N)``.  That is correct only while the register allocator leaves the
destinations alone between the two -- no copy, no spill, no accumulation-register move -- which
nothing in the language promises.  ``violations(asm_text)`` runs a forward data flow over every
kernel's control-flow graph in a gfx950 assembly listing (hipcc --save-temps) and names every
instruction that reads such a destination before an inline-assembly wait.

Used twice: by tests/test_build_guards.py on the library's own translation units, and by
mc3_amd/jit.py on every loop it compiles at run time -- a specialised loop that fails the check is
not installed (the general chain runs that shape instead)."""
import re


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


def violations(text):
    """(kernel, block, instruction, registers) of every read of an in-flight hidden load;
    second value: the number of hidden loads the scan followed."""
    found, checked = [], 0
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
                if touched:
                    found.append((name, lab, f'{op} {rest}', sorted(touched)))
                if not reads_all:         # (overwritten: the allocator holds the value dead)
                    pending -= _regs(rest.partition(',')[0], 'v')
            for nxt in succ[lab]:
                if nxt not in state:
                    continue
                new = frozenset(pending) if state[nxt] is None else state[nxt] | pending
                if new != state[nxt]:
                    state[nxt] = new
                    work.append(nxt)
    return found, checked


def scratch_bytes(text):
    """{kernel name: bytes of scratch per lane} of an assembly listing."""
    out = {}
    meta = text[text.index('amdhsa.kernels:'):]
    for blk in meta.split('  - .agpr_count')[1:]:
        name = re.search(r'\.name:\s+(\S+)', blk).group(1)
        out[name] = int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk).group(1))
    return out
