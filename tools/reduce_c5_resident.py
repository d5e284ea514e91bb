#!/usr/bin/env python3
"""Reduce what tools/pmc_c5.sh gathered (gpurun_out/r02b/) to the tracked files
profiles/r02_c5_resident_* and profiles/r02_mfma_f64_issue_probe.txt."""
import collections
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, 'gpurun_out', 'r02b')
D = os.path.join(ROOT, 'profiles')
shutil.copy(f'{R}/kt_c5/c5_kernel_stats.csv', f'{D}/r02_c5_resident_kernel_stats.csv')
shutil.copy(f'{R}/c5_ticks.txt', f'{D}/r02_c5_resident_ticks.txt')
shutil.copy(f'{R}/c5_bench.jsonl', f'{D}/r02_c5_resident_bench.jsonl')
if os.path.exists(f'{R}/mfma_f64_issue_probe.txt'):
    shutil.copy(f'{R}/mfma_f64_issue_probe.txt', f'{D}/r02_mfma_f64_issue_probe.txt')
out = {'what': 'rocprofv3 --pmc (three separate passes) + --kernel-trace of `python3 '
               'tools/bench_configs.py c5`: resident_tile_kernel<3>, the launch of 300 '
               'iterations (the first launch of 30 is the warm-up)', 'counters': {}}
for n in (1, 2, 3):
    agg = collections.defaultdict(list)
    dur = []
    with open(f'{R}/pmc{n}_c5/c5_counter_collection.csv') as f:
        for r in csv.DictReader(f):
            if 'resident_tile' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
                dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
                out['kernel'] = {k: r[k] for k in ('Kernel_Name', 'Grid_Size', 'Workgroup_Size',
                                                   'Scratch_Size')}
    for k, v in agg.items():
        out['counters'][k] = max(v)
    out[f'pass{n}_duration_ns'] = max(dur)
c, it, waves = out['counters'], 300, 1960
ns = out['pass3_duration_ns']
clk = c['GRBM_GUI_ACTIVE']/8/ns          # GHz (the counter is summed over the 8 XCDs)
simd_cycles = ns*clk*980
out['derived'] = {
    'iterations': it, 'us_per_iteration': ns/it/1e3,
    'shader_clock_GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)': round(clk, 3),
    'matrix_instructions_per_wave_iteration': c['SQ_INSTS_MFMA']/it/waves,
    'valu_instructions_per_wave_iteration': c['SQ_INSTS_VALU']/it/waves,
    'f64_mul_plus_fma_per_wave_iteration':
        (c['SQ_INSTS_VALU_FMA_F64'] + c['SQ_INSTS_VALU_MUL_F64'])/it/waves,
    'salu_per_wave_iteration': c['SQ_INSTS_SALU']/it/waves,
    'lds_per_wave_iteration': c['SQ_INSTS_LDS']/it/waves,
    'vmem_rd_per_wave_iteration': c['SQ_INSTS_VMEM_RD']/it/waves,
    'vmem_wr_per_wave_iteration': c['SQ_INSTS_VMEM_WR']/it/waves,
    'matrix_pipe_busy_fraction (SQ_VALU_MFMA_BUSY_CYCLES / (980 SIMDs x kernel cycles))':
        c['SQ_VALU_MFMA_BUSY_CYCLES']/simd_cycles,
    'wave_cycles_waiting_any': c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES'],
    'wave_cycles_waiting_to_issue': c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES'],
    'wave_cycles_issuing': c['SQ_ACTIVE_INST_ANY']/c['SQ_WAVE_CYCLES'],
    'lds_bank_conflict_cycles': c['SQ_LDS_BANK_CONFLICT'],
    'useful_TFLOPs (2 x 10 coefficients x 1e5 points x 512 chains per iteration)':
        2*10*1e5*512/(ns/it*1e-9)/1e12,
    'issued_matrix_TFLOPs (2048 flops per instruction)':
        c['SQ_INSTS_MFMA']*2048/(ns*1e-9)/1e12,
}
json.dump(out, open(f'{D}/r02_c5_resident_pmc.json', 'w'), indent=1)
print(json.dumps(out['derived'], indent=1))
