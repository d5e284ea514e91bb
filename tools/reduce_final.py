#!/usr/bin/env python3
"""Reduce what tools/collect_final.sh gathered (gpurun_out/r02f/) to tracked files under
profiles/ (the headline trio is written over the earlier r02 files of the same names: same
commands, final library)."""
import collections
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, 'gpurun_out', 'r02f')
D = os.path.join(ROOT, 'profiles')


def last_json_line(path):
    with open(path) as f:
        return json.loads([ln for ln in f.read().splitlines() if ln.startswith('{')][-1])


def counters(path, pick):
    vals = collections.defaultdict(list)
    name = None
    with open(path) as f:
        for r in csv.DictReader(f):
            if pick in r['Kernel_Name']:
                vals[r['Counter_Name']].append(float(r['Counter_Value']))
                name = r['Kernel_Name']
    return vals, name


with open(f'{D}/r02_c2_bench.json', 'w') as f:
    f.write(json.dumps(last_json_line(f'{R}/c2_bench.json')) + '\n')
shutil.copy(f'{R}/kt_c2/c2_kernel_stats.csv', f'{D}/r02_c2_lean_kernel_stats.csv')
shutil.copy(f'{R}/kt_q5/q5_kernel_stats.csv', f'{D}/r02_q5_resident_kernel_stats.csv')
shutil.copy(f'{R}/configs.jsonl', f'{D}/r02_configs_final.jsonl')
shutil.copy(f'{R}/configs_three_launch.jsonl', f'{D}/r02_configs_three_launch.jsonl')
for name in ('e2e_sample_1e7.txt', 'e2e_sample_c5.txt'):
    with open(f'{R}/{name}') as f:
        keep = [ln for ln in f.read().splitlines() if ln.strip() and 'Yippee' not in ln][:24]
    with open(f'{D}/r02_{name}', 'w') as f:
        f.write('\n'.join(keep) + '\n')
fetch, kern = counters(f'{R}/pmc_c2_FETCH_SIZE/c2_counter_collection.csv', 'lean_demc_kernel')
write, _ = counters(f'{R}/pmc_c2_WRITE_SIZE/c2_counter_collection.csv', 'lean_demc_kernel')
fk = sum(fetch['FETCH_SIZE'])/len(fetch['FETCH_SIZE'])
wk = sum(write['WRITE_SIZE'])/len(write['WRITE_SIZE'])
steps = 16384*256
rec = {
    'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, + --kernel-trace for '
              'the kernel names), python3 bench.py --no-cpu-baseline; tools/collect_final.sh',
    'kernel': kern.replace('void ', '').split('(')[0],
    'refreshes_per_launch': 16384, 'chains': 256, 'launches_averaged': len(fetch['FETCH_SIZE']),
    'FETCH_SIZE_KB_per_launch': fk, 'WRITE_SIZE_KB_per_launch': wk,
    'correction': 'MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reads 1/2 of the bytes of a '
                  'wide coalesced stream, so the read side is doubled; these reads are narrow (8 B '
                  'per lane), an uncalibrated width, so the doubled figure is an upper bound. '
                  'WRITE_SIZE is taken as is.',
    'hbm_bytes_per_chain_step': (2*fk + wk)*1024/steps,
    'hbm_bytes_per_chain_step_uncorrected': (fk + wk)*1024/steps,
}
with open(f'{D}/r02_c2_pmc.json', 'w') as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec, indent=1))
print(open(f'{D}/r02_c2_bench.json').read()[:300])
