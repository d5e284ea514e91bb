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
        if op == 's_branch' or op.startswith('s_cbranch') or op == 's_endpgm':
            # a block ends at its branch: what follows (the compiler marks it `; %bb.N:`, a
            # comment, not a label) is a block of its own -- the state handed to the branch's
            # target must be the state AT the branch, not at the end of the text behind it
            synth = f'{label}+{len(order) + 1}'
            close(synth)
            label, cur = synth, []
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


_VMEM = ('global_load', 'global_store', 'global_atomic', 'buffer_load', 'buffer_store',
         'buffer_atomic', 'flat_load', 'flat_store', 'flat_atomic', 'scratch_load',
         'scratch_store')
_KCAP = 64      # (younger-operation counts saturate here: the counter has six bits)


def _vmcnt(rest):
    """The N of ``s_waitcnt ... vmcnt(N) ...`` (None: the instruction does not wait on the
    vector-memory counter)."""
    m = re.search(r'vmcnt\((\d+)\)', rest)
    if m:
        return int(m.group(1))
    m = re.match(r'\s*(0x[0-9a-fA-F]+|\d+)\s*$', rest)
    if m:                       # a raw immediate: vmcnt = bits 3:0 and 15:14 (gfx9)
        imm = int(m.group(1), 0)
        n = (imm & 0xf) | ((imm >> 14) & 0x3) << 4
        return None if n == 63 else n
    return None


def violations(text):
    """(kernel, block, instruction, registers) of every read of an in-flight hidden load;
    second value: the number of hidden loads the scan followed.

    The vector-memory counter retires loads AND stores in issue order (gfx9), and
    ``s_waitcnt vmcnt(N)`` only says "at most N still in flight": a hidden load is known to
    have landed only if at least N vector-memory operations were issued AFTER it.  So the state
    is {destination register: k}, k = the least number of vector-memory operations (of any
    kind, the compiler's own included) issued after the load on any path to this point;
    ``vmcnt(N)`` -- hand-written or the compiler's -- retires the registers with k >= N and
    leaves the younger ones pending (round 4 treated any inline-assembly wait as a full one,
    which was right only while the one operation left outstanding at a ``vmcnt(1)`` site was
    a store)."""
    found, checked = [], 0
    for name, body in _kernels(text):
        blocks, succ, order = _blocks(body)
        state = {lab: None for lab in order}        # {register: k} at block entry
        index = {lab: i for i, lab in enumerate(order)}
        state[order[0]] = {}
        work = [order[0]]
        while work:
            lab = work.pop()
            pending = dict(state[lab])
            for inasm, op, rest in blocks[lab]:
                if op == 's_waitcnt':
                    n = _vmcnt(rest)
                    if n is not None:
                        pending = {r: k for r, k in pending.items() if k < n}
                    continue
                is_vmem = op.startswith(_VMEM)
                hidden = inasm and op == 'global_load_dwordx2' \
                    and not rest.lstrip().startswith('a[')
                if not inasm or (is_vmem and not hidden):
                    # what the instruction READS: every operand of a store / compare / atomic,
                    # the operands behind the destination otherwise.  (Writes are not flagged:
                    # the structurised control flow has statically possible, dynamically dead
                    # paths -- a "break" flag routed through the loop latch -- on which the
                    # allocator rightly reuses a register whose value is dead there.)
                    reads_all = op.startswith(('global_store', 'scratch_store', 'buffer_store',
                                               'flat_store', 'ds_write', 'ds_store', 'v_cmp',
                                               'global_atomic', 'flat_atomic'))
                    srcs = rest if reads_all else rest.partition(',')[2]
                    touched = _regs(srcs, 'v') & set(pending)
                    if touched and not inasm:
                        found.append((name, lab, f'{op} {rest}', sorted(touched)))
                    if not reads_all and not inasm:   # (overwritten: the value is held dead)
                        for r in _regs(rest.partition(',')[0], 'v'):
                            pending.pop(r, None)
                if is_vmem:
                    pending = {r: min(k + 1, _KCAP) for r, k in pending.items()}
                    if hidden:
                        for r in _regs(rest.split(',')[0], 'v'):
                            pending[r] = 0
                        checked += 1
            for nxt in succ[lab]:
                if nxt not in state:
                    continue
                if pending and index[nxt] <= index[lab]:
                    # A BACKWARD jump with loads still pending.  The loops ask at the top of an
                    # iteration and wait at its bottom, so the only way to the latch without
                    # the wait is the way OUT (a wait for a row gave up: `goto out`), which the
                    # structuriser routes through the latch behind a flag -- statically a path
                    # into the next iteration, dynamically the loop's exit.  Followed, it makes
                    # every scratch use of such a register at the top of the body (the refill
                    # of a batch of draws) a "read of an in-flight load".  Not followed; what
                    # the scan then does NOT see: a reuse on that exit path itself, behind the
                    # loop, where nothing reads Z rows any more.
                    continue
                if state[nxt] is None:
                    new = dict(pending)
                else:
                    new = dict(state[nxt])
                    for r, k in pending.items():
                        new[r] = min(k, new.get(r, _KCAP))
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
