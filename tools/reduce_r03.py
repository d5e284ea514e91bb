#!/usr/bin/env python3
"""Reduce what tools/collect_r03.sh gathered (gpurun_out/r03/) to the small tracked files
profiles/r03_*: bench lines, rocprofv3 kernel-stats CSVs, PMC summaries."""
import collections
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, 'gpurun_out', 'r03')
D = os.path.join(ROOT, 'profiles')


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


def have(rel):
    p = os.path.join(R, rel)
    return os.path.exists(p) and os.path.getsize(p) > 0


for rel in ('c2_bench.json', 'c2_bench_2ranks_on_1gpu.json', 'c3_bench_8ranks_on_1gpu.json',
            'c5_bench_4ranks_on_1gpu.json', 'c2_bench_rccl_graph_1rank.json'):
    if have(rel):
        line = last_json_line(os.path.join(R, rel))
        if line:
            with open(os.path.join(D, 'r03_' + rel), 'w') as f:
                f.write(json.dumps(line) + '\n')
for rel, name in (('configs.jsonl', 'configs.jsonl'),
                  ('configs_three_launch.jsonl', 'configs_three_launch.jsonl'),
                  ('configs_no_chainwave.jsonl', 'configs_no_chainwave.jsonl'),
                  ('configs_general_chain.jsonl', 'configs_general_chain.jsonl'),
                  ('kt_c2/c2_kernel_stats.csv', 'c2_lean_kernel_stats.csv'),
                  ('kt_c2/c2_agent_info.csv', 'agent_info.csv'),
                  ('kt_c4/c4_kernel_stats.csv', 'c4_wavelet_kernel_stats.csv'),
                  ('kt_c5/c5_kernel_stats.csv', 'c5_resident_kernel_stats.csv'),
                  ('kt_c2snooker/c2snooker_kernel_stats.csv', 'c2snooker_kernel_stats.csv'),
                  ('kt_c2mrw/c2mrw_kernel_stats.csv', 'c2mrw_kernel_stats.csv'),
                  ('kt_c3x2snooker/c3x2snooker_kernel_stats.csv',
                   'c3x2snooker_kernel_stats.csv'),
                  ('kt_c2thin10demc/c2thin10demc_kernel_stats.csv',
                   'c2thin10demc_kernel_stats.csv'),
                  ('handoff_probe.json', 'handoff_probe_one_gpu.json')):
    if have(rel):
        shutil.copy(os.path.join(R, rel), os.path.join(D, 'r03_' + name))
if have('soak_demc_1e6.txt'):
    with open(os.path.join(R, 'soak_demc_1e6.txt')) as f:
        keep = [ln for ln in f.read().splitlines() if ln.strip() and 'Yippee' not in ln][:22]
    with open(os.path.join(D, 'r03_soak_demc_1e6.txt'), 'w') as f:
        f.write('\n'.join(keep) + '\n')

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
                  'tools/collect_r03.sh',
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
    with open(os.path.join(D, 'r03_c2_pmc.json'), 'w') as f:
        json.dump(rec, f, indent=1)

# C4 / C5: instruction mix, stalls, matrix pipe of the dominant kernel (three passes)
for cfg, pick, iters, out_name in (('c4', 'wavelet_cw_kernel', 4000, 'r03_c4_pmc.json'),
                                   ('c5', 'resident_tile_kernel', 300, 'r03_c5_resident_pmc.json')):
    if not have(f'pmc1_{cfg}/{cfg}_counter_collection.csv'):
        continue
    out = {'what': f'rocprofv3 --pmc (three separate passes) + --kernel-trace of `python3 '
                   f'tools/bench_configs.py {cfg}`: the launch of {iters} iterations (the first, '
                   'shorter launch is the warm-up); tools/collect_r03.sh', 'counters': {}}
    for n in (1, 2, 3):
        p = os.path.join(R, f'pmc{n}_{cfg}/{cfg}_counter_collection.csv')
        if not os.path.exists(p):
            continue
        vals, meta, dur = counters(p, pick)
        out['kernel'] = meta
        for k, v in vals.items():
            out['counters'][k] = max(v)                # (the long launch)
        out[f'pass{n}_duration_ns'] = max(dur)
    c = out['counters']
    waves = c['SQ_WAVES']
    wc = c['SQ_WAVE_CYCLES']
    d = {'iterations': iters, 'waves': waves,
         'us_per_iteration (under the profiler)': out['pass1_duration_ns']/iters/1e3,
         'valu_per_wave_iteration': c['SQ_INSTS_VALU']/iters/waves,
         'salu_per_wave_iteration': c['SQ_INSTS_SALU']/iters/waves,
         'lds_per_wave_iteration': c['SQ_INSTS_LDS']/iters/waves,
         'vmem_rd_per_wave_iteration': c['SQ_INSTS_VMEM_RD']/iters/waves,
         'vmem_wr_per_wave_iteration': c['SQ_INSTS_VMEM_WR']/iters/waves,
         'wave_cycles_issuing (SQ_ACTIVE_INST_ANY)': c['SQ_ACTIVE_INST_ANY']/wc,
         'wave_cycles_parked (SQ_WAIT_ANY: s_waitcnt / barrier)': c['SQ_WAIT_ANY']/wc,
         'wave_cycles_waiting_to_issue (SQ_WAIT_INST_ANY)': c['SQ_WAIT_INST_ANY']/wc,
         'lds_bank_conflict_cycles': c.get('SQ_LDS_BANK_CONFLICT')}
    if 'GRBM_GUI_ACTIVE' in c:
        ns = out['pass3_duration_ns']
        clk = c['GRBM_GUI_ACTIVE']/8/ns
        d['shader_clock_GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)'] = round(clk, 3)
        d['f64_mul_plus_fma_per_wave_iteration'] = \
            (c['SQ_INSTS_VALU_FMA_F64'] + c['SQ_INSTS_VALU_MUL_F64'])/iters/waves
        if c.get('SQ_INSTS_MFMA'):
            simds = waves/2                            # two waves per SIMD
            d['matrix_instructions_per_wave_iteration'] = c['SQ_INSTS_MFMA']/iters/waves
            d['matrix_pipe_busy_fraction (SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles))'] = \
                c['SQ_VALU_MFMA_BUSY_CYCLES']/(ns*clk*simds)
            d['useful_TFLOPs (2 x 10 coefficients x 1e5 points x 512 chains per iteration)'] = \
                2*10*1e5*512/(ns/iters*1e-9)/1e12
            d['issued_matrix_TFLOPs (2048 flops per instruction)'] = \
                c['SQ_INSTS_MFMA']*2048/(ns*1e-9)/1e12
    out['derived'] = d
    with open(os.path.join(D, out_name), 'w') as f:
        json.dump(out, f, indent=1)
    print(out_name, json.dumps(d, indent=1))
print(sorted(p for p in os.listdir(D) if p.startswith('r03')))
