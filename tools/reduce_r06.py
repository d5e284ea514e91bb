#!/usr/bin/env python3
"""Reduce what tools/collect_r06.sh gathered (gpurun_out/r06/) to the small tracked files
profiles/r06_*: bench lines, configuration tables, the delay sweep, rocprofv3 kernel-stats
CSVs, PMC summaries."""
import collections
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, 'gpurun_out', 'r06')
D = os.path.join(ROOT, 'profiles')


def have(rel):
    p = os.path.join(R, rel)
    return os.path.exists(p) and os.path.getsize(p) > 0


def last_json_line(path):
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.startswith('{')]
    return json.loads(lines[-1]) if lines else None


def counters(path, pick):
    vals = collections.defaultdict(list)
    meta, dur = {}, []
    with open(path) as f:
        for r in csv.DictReader(f):
            if pick in r['Kernel_Name']:
                vals[r['Counter_Name']].append(float(r['Counter_Value']))
                meta = {k: r[k] for k in ('Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'VGPR_Count',
                                          'Accum_VGPR_Count', 'SGPR_Count', 'Scratch_Size',
                                          'LDS_Block_Size')}
                dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    return vals, meta, dur


for rel in ('c2_bench.json', 'c2_bench_2ranks_on_1gpu.json', 'c3_bench_8ranks_on_1gpu.json',
            'c3_bench_8ranks_on_1gpu_strict.json', 'c5_bench_4ranks_on_1gpu.json'):
    if have(rel):
        line = last_json_line(os.path.join(R, rel))
        if line:
            with open(os.path.join(D, 'r06_' + rel), 'w') as f:
                f.write(json.dumps(line) + '\n')
for rel, name in (('configs.jsonl', 'configs.jsonl'),
                  ('offlean_configs.jsonl', 'offlean_configs.jsonl'),
                  ('offlean_configs_no_lean.jsonl', 'offlean_configs_before.jsonl'),
                  ('lag_configs.jsonl', 'lag_configs.jsonl'),
                  ('delay_sweep.jsonl', 'delay_sweep.jsonl'),
                  ('delay_sweep_c3.jsonl', 'delay_sweep_c3.jsonl'),
                  ('delay_sweep_c3big.jsonl', 'delay_sweep_c3big.jsonl'),
                  ('kt_c2/c2_kernel_stats.csv', 'c2_lean_kernel_stats.csv'),
                  ('kt_c2/c2_agent_info.csv', 'agent_info.csv'),
                  ('kt_c3x4/c3x4_kernel_stats.csv', 'c3x4_kernel_stats.csv'),
                  ('kt_c4/c4_kernel_stats.csv', 'c4_wavelet_kernel_stats.csv'),
                  ('kt_c4x2/c4x2_kernel_stats.csv', 'c4x2_wavelet_kernel_stats.csv'),
                  ('c4_pipe_diag.txt', 'c4_pipe_diag.txt'),
                  ('hbm_read_probe.json', 'hbm_read_probe.json'),
                  ('mid_range.jsonl', 'mid_range.jsonl'),
                  ('kt_c3one/c3one_kernel_stats.csv', 'c3one_kernel_stats.csv'),
                  ('kt_c5/c5_kernel_stats.csv', 'c5_resident_kernel_stats.csv'),
                  ('e2e_sample.txt', 'e2e_sample.txt')):
    if have(rel):
        shutil.copy(os.path.join(R, rel), os.path.join(D, 'r06_' + name))

# the wavelet configurations measured again after the pipelined kernel: their lines replace the
# earlier ones in r06_configs.jsonl (which move to r06_configs_c4_before.jsonl)
if have('c4family_configs.jsonl') and not os.path.exists(os.path.join(D, 'r06_configs.jsonl')):
    shutil.copy(os.path.join(R, 'c4family_configs.jsonl'), os.path.join(D, 'r06_configs.jsonl'))
elif have('c4family_configs.jsonl'):
    new = {}
    for line in open(os.path.join(R, 'c4family_configs.jsonl')):
        if line.startswith('{'):
            new[json.loads(line)['config']] = line
    old = open(os.path.join(D, 'r06_configs.jsonl')).read().splitlines(True)
    before_path = os.path.join(D, 'r06_configs_c4_before.jsonl')
    keep, before = [], []
    for line in old:
        name = json.loads(line)['config']
        if name in new:
            before.append(line)
            keep.append(new.pop(name))
        else:
            keep.append(line)
    keep += list(new.values())
    if before and not os.path.exists(before_path):
        open(before_path, 'w').writelines(before)
    open(os.path.join(D, 'r06_configs.jsonl'), 'w').writelines(keep)

# C2: HBM traffic of the headline kernel (one counter per pass)
if have('pmc_c2_FETCH_SIZE/c2_counter_collection.csv'):
    fetch, meta, _ = counters(os.path.join(R, 'pmc_c2_FETCH_SIZE/c2_counter_collection.csv'),
                              'lean_demc_kernel')
    write, _, _ = counters(os.path.join(R, 'pmc_c2_WRITE_SIZE/c2_counter_collection.csv'),
                           'lean_demc_kernel')
    fk = sum(fetch['FETCH_SIZE'])/len(fetch['FETCH_SIZE'])
    wk = sum(write['WRITE_SIZE'])/len(write['WRITE_SIZE'])
    steps = 16384*256
    rec = {
        'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, + --kernel-trace '
                  'for the kernel names), python3 bench.py --no-cpu-baseline --no-e2e; '
                  'tools/collect_r06.sh',
        'kernel': meta['Kernel_Name'].replace('void ', '').split('(')[0],
        'refreshes_per_launch': 16384, 'chains': 256,
        'launches_averaged': len(fetch['FETCH_SIZE']),
        'FETCH_SIZE_KB_per_launch': fk, 'WRITE_SIZE_KB_per_launch': wk,
        'correction': 'MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reads 1/2 of the bytes '
                      'of a wide coalesced stream, so the read side is doubled; these reads are '
                      'narrow (8 B per lane), an uncalibrated width, so the doubled figure is an '
                      'upper bound.  WRITE_SIZE is taken as is.',
        'hbm_bytes_per_chain_step': (2*fk + wk)*1024/steps,
        'hbm_bytes_per_chain_step_uncorrected': (fk + wk)*1024/steps,
    }
    with open(os.path.join(D, 'r06_c2_pmc.json'), 'w') as f:
        json.dump(rec, f, indent=1)

# large ensembles: instruction mix and stall counters of the several-chains-per-wave loops
BIG = {'c3x4': ('lean_demc_rows_kernel', 2000, 8192), 'c3x2mrw': ('lean_mrw_mc_kernel', 3000, 4096),
       # (round 6: the child processes of the bench line -- launches of 16 384 refreshes, C5's
       # ONE launch of 8 192)
       'c3one': ('lean_demc_kernel', 16384, 2048),
       'c4': ('wavelet_pipe_kernel', 16384, 256), 'c4x2': ('wavelet_pipe_kernel', 2000, 512),
       'c5': ('resident_async_kernel', 8192, 512)}
CHILD = ('c3one', 'c4', 'c5')
for cfg, (pick, iters, chains) in BIG.items():
    if not have(f'pmc1_{cfg}/{cfg}_counter_collection.csv'):
        continue
    cmd = f'bench.py --config-child {cfg}' if cfg in CHILD else f'tools/bench_configs.py {cfg}'
    out = {'what': f'rocprofv3 --pmc (three separate passes) + --kernel-trace of `python3 '
                   f'{cmd}`: the launch of {iters} refreshes of {chains} '
                   'chains (the first, shorter launch is the warm-up); tools/collect_r06.sh',
           'counters': {}}
    for n in (1, 2, 3):
        p = os.path.join(R, f'pmc{n}_{cfg}/{cfg}_counter_collection.csv')
        if not os.path.exists(p):
            continue
        vals, meta, dur = counters(p, pick)
        if not vals:
            continue
        out['kernel'] = meta
        for k, v in vals.items():
            out['counters'][k] = max(v)                # (the long launch)
        out[f'pass{n}_duration_ns'] = max(dur)
    c = out['counters']
    if 'SQ_WAVES' not in c or 'SQ_WAVE_CYCLES' not in c:
        continue
    waves, wc = c['SQ_WAVES'], c['SQ_WAVE_CYCLES']
    steps = iters*chains
    ns = out.get('pass1_duration_ns', 0)
    d = {'refreshes': iters, 'chains': chains, 'waves': waves,
         'us_per_refresh (under the profiler)': ns/iters/1e3,
         'million_chain_steps_per_s (under the profiler)': steps/(ns*1e-9)/1e6 if ns else None,
         'valu_per_chain_step': c['SQ_INSTS_VALU']/steps,
         'salu_per_chain_step': c['SQ_INSTS_SALU']/steps,
         'lds_per_chain_step': c.get('SQ_INSTS_LDS', 0)/steps,
         'vmem_per_chain_step': (c.get('SQ_INSTS_VMEM_RD', 0) + c.get('SQ_INSTS_VMEM_WR', 0))/steps,
         'wave_cycles_issuing (SQ_ACTIVE_INST_ANY)': c['SQ_ACTIVE_INST_ANY']/wc,
         'wave_cycles_parked (SQ_WAIT_ANY)': c['SQ_WAIT_ANY']/wc,
         'wave_cycles_waiting_to_issue (SQ_WAIT_INST_ANY)': c['SQ_WAIT_INST_ANY']/wc,
         'SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES': c['SQ_ACTIVE_INST_VALU']/c['SQ_BUSY_CYCLES']}
    f64 = sum(c.get(k, 0) for k in ('SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_MUL_F64',
                                     'SQ_INSTS_VALU_ADD_F64'))
    if f64:
        d['f64_fma_mul_add_per_chain_step'] = f64/steps
        if ns:
            # FP64 vector issue: 4 cycles per wave instruction, 1024 SIMDs at the measured clock
            clk = c['GRBM_GUI_ACTIVE']/8/out['pass3_duration_ns'] if 'GRBM_GUI_ACTIVE' in c else 2.4
            d['shader_clock_GHz'] = round(clk, 3)
            d['fp64_issue_frac'] = f64*4/(1024*clk*ns)
            d['valu_issue_frac (all vector instructions x 4 cycles)'] = \
                c['SQ_INSTS_VALU']*4/(1024*clk*ns)
    out['derived'] = d
    name = f'r06_{cfg}_pmc.json'
    with open(os.path.join(D, name), 'w') as f:
        json.dump(out, f, indent=1)
    print(name, json.dumps(d, indent=1))
print(sorted(p for p in os.listdir(D) if p.startswith('r06')))

if have('gpu_tests.log'):
    with open(os.path.join(R, 'gpu_tests.log')) as f:
        tail = [ln for ln in f.read().splitlines() if ' passed' in ln or ' failed' in ln]
    with open(os.path.join(D, 'r06_gpu_tests.txt'), 'w') as f:
        f.write('python -m pytest tests -m gpu -x -q   (tools/collect_r06.sh tests, one MI355X)\n'
                + '\n'.join(tail) + '\n')
