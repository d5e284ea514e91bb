#!/usr/bin/env python3
"""Reduce what tools/collect_profiles.sh gathered (gpurun_out/<tag>/) to the small tracked
files under profiles/: bench lines, rocprofv3 kernel-stats CSVs, PMC summaries.

    python tools/reduce_profiles.py r02
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
            meta[r['Kernel_Name']] = {k: r[k] for k in ('Grid_Size', 'Workgroup_Size', 'VGPR_Count',
                                                        'Accum_VGPR_Count', 'SGPR_Count',
                                                        'Scratch_Size', 'LDS_Block_Size')}
    return agg, meta


def last_json_line(path):
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.startswith('{')]
    return json.loads(lines[-1])


def copy(rel, name):
    shutil.copy(os.path.join(src, rel), os.path.join(dst, f'{tag}_{name}'))


# bench lines
for rel, name in (('c2_bench.json', 'c2_bench.json'),
                  ('c2_bench_rccl_graph_1rank.json', 'c2_bench_rccl_graph_1rank.json'),
                  ('c2_bench_2ranks_on_1gpu.json', 'c2_bench_2ranks_on_1gpu.json'),
                  ('c2_bench_8ranks_on_1gpu.json', 'c2_bench_8ranks_on_1gpu.json')):
    p = os.path.join(src, rel)
    if os.path.exists(p) and os.path.getsize(p):
        with open(os.path.join(dst, f'{tag}_{name}'), 'w') as f:
            f.write(json.dumps(last_json_line(p)) + '\n')
copy('configs.jsonl', 'configs.jsonl')
copy('kt_c2/c2_kernel_stats.csv', 'c2_lean_kernel_stats.csv')
copy('kt_c2/c2_agent_info.csv', 'agent_info.csv')
copy('kt_c4/c4_kernel_stats.csv', 'c4_wavelet_kernel_stats.csv')
copy('kt_c5/c5_kernel_stats.csv', 'c5_tiled_kernel_stats.csv')
for name in ('e2e_sample_1e7.txt', 'soak_demc_1e6.txt'):
    with open(os.path.join(src, name)) as f:
        keep = [ln for ln in f.read().splitlines() if ln.strip() and 'Yippee' not in ln][:22]
    with open(os.path.join(dst, f'{tag}_{name}'), 'w') as f:
        f.write('\n'.join(keep) + '\n')

# C2: HBM traffic of the persistent kernel (one counter per pass)
fetch, _ = counters(os.path.join(src, 'pmc_c2_FETCH_SIZE', 'c2_counter_collection.csv'))
write, _ = counters(os.path.join(src, 'pmc_c2_WRITE_SIZE', 'c2_counter_collection.csv'))
kern = next(k for k in fetch if 'lean_demc_kernel' in k)
fk = sum(fetch[kern]['FETCH_SIZE'])/len(fetch[kern]['FETCH_SIZE'])
wk = sum(write[kern]['WRITE_SIZE'])/len(write[kern]['WRITE_SIZE'])
steps = 16384*256
rec = {
    'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, + --kernel-trace for '
              'the kernel names), python3 bench.py --no-cpu-baseline; tools/collect_profiles.sh',
    'kernel': kern.replace('void ', '').split('(')[0],
    'refreshes_per_launch': 16384, 'chains': 256, 'launches_averaged': len(fetch[kern]['FETCH_SIZE']),
    'FETCH_SIZE_KB_per_launch': fk, 'WRITE_SIZE_KB_per_launch': wk,
    'correction': 'MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reads 1/2 of the bytes of a '
                  'wide coalesced stream, so the read side is doubled; these reads are narrow (8 B '
                  'per lane), an uncalibrated width, so the doubled figure is an upper bound. '
                  'WRITE_SIZE is taken as is.',
    'hbm_bytes_per_chain_step': (2*fk + wk)*1024/steps,
    'hbm_bytes_per_chain_step_uncorrected': (fk + wk)*1024/steps,
}
with open(os.path.join(dst, f'{tag}_c2_pmc.json'), 'w') as f:
    json.dump(rec, f, indent=1)

# C4 / C5: instruction mix and stall counters of the dominant kernel
for cfg, pick, iters in (('c4', 'persist_kernel', 4400), ('c5', 'tile_chisq', None)):
    out = {'command': f'rocprofv3 --pmc <8 SQ counters> --kernel-trace --output-format csv -- '
                      f'python3 tools/bench_configs.py {cfg} (two passes, tools/collect_profiles.sh)'}
    tot = {}
    for p in ('pmc1', 'pmc2'):
        agg, meta = counters(os.path.join(src, f'{p}_{cfg}', f'{cfg}_counter_collection.csv'))
        kern = next(k for k in agg if pick in k)
        out['kernel'] = kern
        out.update(meta[kern])
        for c, v in agg[kern].items():
            tot[c] = sum(v)
            out.setdefault('launches', len(v))
    out['sum_over_launches'] = tot
    waves = tot['SQ_WAVES']
    if iters:
        # SQ_WAVES is summed over the launches; the same waves run every iteration of a launch
        wi = waves/out['launches']*iters
        out['iterations_total'] = iters
        out['per_wave_iteration'] = {c: tot[c]/wi for c in
                                     ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS',
                                      'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR')}
        n = sum(out['per_wave_iteration'].values())
        out['salu_share_of_instructions'] = out['per_wave_iteration']['SQ_INSTS_SALU']/n
    wc = tot['SQ_WAVE_CYCLES']
    out['share_of_wave_cycles'] = {
        'issuing (SQ_ACTIVE_INST_ANY)': tot['SQ_ACTIVE_INST_ANY']/wc,
        'parked at s_waitcnt / barrier (SQ_WAIT_ANY)': tot['SQ_WAIT_ANY']/wc,
        'issue stall (SQ_WAIT_INST_ANY)': tot['SQ_WAIT_INST_ANY']/wc,
        'of which LDS (SQ_WAIT_INST_LDS)': tot['SQ_WAIT_INST_LDS']/wc}
    if tot.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        # MFMA_BUSY counts cycles, WAVE_CYCLES quad-cycles per wave
        out['mfma_busy_cycles_per_launch'] = tot['SQ_VALU_MFMA_BUSY_CYCLES']/out['launches']
    with open(os.path.join(dst, f'{tag}_{cfg}_pmc.json'), 'w') as f:
        json.dump(out, f, indent=1)
print(sorted(p for p in os.listdir(dst) if p.startswith(tag)))
