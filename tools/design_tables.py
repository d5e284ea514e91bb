#!/usr/bin/env python3
"""Markdown tables for DESIGN.md from the committed profiles (off-lean shapes; delay sweep)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')


def rows(name):
    path = os.path.join(P, name)
    return [json.loads(l) for l in open(path)] if os.path.exists(path) else []


def offlean():
    now = {r['config']: r for r in rows('r04_offlean_configs.jsonl')}
    before = {r['config']: r for r in rows('r04_offlean_configs_before.jsonl')}
    lean = {'demc': 217.0, 'snooker': 316.0, 'mrw': 438.0}
    what = {
        'off_shared': 'one shared parameter (demc)', 'off_eps': 'fepsilon = 0.01 (demc)',
        'off_eps_snooker': 'fepsilon = 0.01 (snooker)',
        'off_poly5': 'polynomial, 5 coefficients', 'off_poly8': 'polynomial, 8 coefficients',
        'off_lin5': 'design matrix, 5 columns', 'off_lin10': 'design matrix, 10 columns',
        'off_prior_thin10': 'priors, thinning 10 (demc)', 'off_prior_snooker': 'snooker with a prior',
        'off_prior': 'priors (demc)', 'off_shared_thin10': 'shared parameter, thinning 10',
        'off_poly5_snooker': 'polynomial 5, snooker', 'off_lin10_snooker': 'design matrix 10, snooker'}
    print('| shape | before | now | of the lean rate of its sampler | kernel now |')
    print('|---|---|---|---|---|')
    for k, r in now.items():
        b = before.get(k, {}).get('million_chain_steps_per_s', float('nan'))
        v = r['million_chain_steps_per_s']
        plan = r['plan']
        kern = ('run-time specialised ' if plan.get('jit') else '') + \
            {2: 'hand-laid-out loop', 1: 'lean `run_chain`', 0: 'general `run_chain`'}[plan['fast']]
        print(f"| {what.get(k, k)} | {b:.0f} | **{v:.0f}** | {v/lean[r['sampler']]:.2f} | {kern} |")


def sweep():
    last = {}
    for r in rows('r04_delay_sweep.jsonl'):
        if 'value' in r:
            last[(r['ranks'], r['zlag'], r['delay_ns'])] = r
    delays = sorted({k[2] for k in last})
    print('| ranks | cadence | ' + ' | '.join(f'{d/1000:g} us' for d in delays) + ' | at 2 us / at 0 |')
    print('|---|---|' + '---|'*(len(delays) + 1))
    for n in sorted({k[0] for k in last}):
        for z in sorted({k[1] for k in last if k[0] == n}):
            vals = [last.get((n, z, d), {}).get('value') for d in delays]
            cells = ' | '.join('—' if v is None else f'{v:.0f}' for v in vals)
            v0, v2 = last.get((n, z, 0), {}).get('value'), last.get((n, z, 2000), {}).get('value')
            ratio = f'{v2/v0:.2f}' if v0 and v2 else '—'
            cad = 'strict (zlag 0)' if z == 0 else f'zlag {z}'
            print(f'| {n} | {cad} | {cells} | {ratio} |')


if __name__ == '__main__':
    offlean()
    print()
    sweep()
