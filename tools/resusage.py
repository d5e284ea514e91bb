#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel
(demangled name, VGPR, AGPR, SGPR, scratch bytes/lane, waves/SIMD, LDS bytes).

    python -m mc3_amd.build --resource-usage > /tmp/res.txt      # every translation unit
    python tools/resusage.py /tmp/res.txt [substring ...]
"""
import re
import shutil
import subprocess
import sys

txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
filt = shutil.which('c++filt') or shutil.which('llvm-cxxfilt')
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].strip()

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    name = name.split(' [')[0]
    if filt:
        name = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip()
    else:
        # no demangler in the image: kernel name + its integer / bool template arguments
        m = re.match(r'_ZN6mc3hip\d+(\w+?)(?:I((?:L[ibj]\d+E)+)E)?(?:v|E)', name)
        if m:
            args = re.findall(r'L([ibj])(\d+)E', m.group(2) or '')
            name = m.group(1) + ('<' + ', '.join(
                ('true' if v == '1' else 'false') if k == 'b' else v for k, v in args) + '>'
                if args else '')
    name = name.replace('mc3hip::', '').replace('void ', '')
    if pats and not all(p in name for p in pats):
        continue
    scratch = g(r'ScratchSize \[bytes/lane\]')
    occ = g(r'Occupancy \[waves/SIMD\]')
    lds = g(r'LDS Size \[bytes/block\]')
    print(f"{name[:100]:100s} vgpr {g('VGPRs'):3d} agpr {g('AGPRs'):3d} sgpr {g('SGPRs'):3d} "
          f"scratch {scratch:4d} occ {occ} lds {lds}")
