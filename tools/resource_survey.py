#!/usr/bin/env python3
"""Development aid: registers, spills and scratch of every kernel of the library (each
translation unit compiled once more with --save-temps, in parallel; a few minutes).
    python tools/resource_survey.py            # summary + the kernels that carry scratch"""
import concurrent.futures
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'mc3_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
         '-fvisibility=hidden', '--save-temps']


def survey(unit, tmp):
    d = os.path.join(tmp, os.path.basename(unit)[:-4])
    os.makedirs(d)
    subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-c', unit, '-o', '/dev/null'], cwd=d,
                   check=True, capture_output=True)
    rows = []
    for f in glob.glob(os.path.join(d, '*gfx950.s')):
        t = open(f).read()
        if 'amdhsa.kernels:' not in t:
            continue
        for blk in t[t.index('amdhsa.kernels:'):].split('  - .agpr_count')[1:]:
            g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))  # noqa: E731
            name = re.search(r'\.name:\s+(\S+)', blk).group(1)
            demangled = subprocess.run(['c++filt', name], capture_output=True,
                                       text=True).stdout.strip()
            rows.append((os.path.basename(unit), demangled.split('(')[0][:90],
                         g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'),
                         g('private_segment_fixed_size')))
    return rows


def main():
    units = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    with tempfile.TemporaryDirectory() as tmp, \
            concurrent.futures.ThreadPoolExecutor(os.cpu_count() or 4) as pool:
        rows = [r for part in pool.map(lambda u: survey(u, tmp), units) for r in part]
    scratch = [r for r in rows if r[5] > 0]
    print(f'{len(rows)} kernels in {len(units)} units; {len(scratch)} carry scratch')
    for r in sorted(scratch, key=lambda r: -r[5]):
        print(f'  {r[5]:5d} B scratch, {r[3]:3d} VGPRs spilled, {r[4]:3d} SGPRs spilled  '
              f'{r[0]}: {r[1]}')


if __name__ == '__main__':
    main()
