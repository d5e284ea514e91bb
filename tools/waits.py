#!/usr/bin/env python3
"""Development aid: the vector-memory operations and s_waitcnt vmcnt(...) of one kernel in a
gfx950 assembly file (hipcc --save-temps), with the basic block each sits in -- to see where the
compiler guards a loop head or a join with a wait for everything in flight.
    python tools/waits.py file.s 'lean_mrw_kernelILi3ELb0E'"""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_ZN6mc3hip\d*' + re.escape(pat) + r'.*:', l))
label, loop = '', ''
for i in range(start, len(lines)):
    l = lines[i]
    if l.startswith('.Lfunc_end'):
        break
    if l.startswith('.LBB'):
        label = l.split(':')[0]
        loop = 'loop' if 'Loop' in l else ''
        continue
    if 'Loop' in l and l.strip().startswith(';'):
        loop = 'loop'
    if re.search(r's_waitcnt.*vmcnt|global_(load|store|atomic)|scratch_|flat_(load|store)', l):
        print(f'{i - start:6d} {label:12s} {loop:5s} {l.strip()[:100]}')
