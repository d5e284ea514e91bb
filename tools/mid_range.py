#!/usr/bin/env python3
"""The sweep behind VERDICT r4 #5: what an iteration costs across the boundaries between the
kernel families -- one register tile / two / streamed tiles / the one-launch tiled plan for the
Gaussian likelihood, one wave / four / eight(+3) waves for the wavelet likelihood -- and the
shapes DESIGN.md section 5 lists as "falls to the general chain".  One JSON line per shape:
the plan the library picked, us per iteration, M chain-steps/s and the PER-POINT rate
(chains x points per second), the figure neighbours are compared by: a shape below 0.5x the
per-point rate of BOTH its neighbours in the sweep is marked `"cliff": true`.

    python tools/mid_range.py [gauss|wavelet|offlean|wlone ...]  > profiles/rNN_mid_range.jsonl
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import helpers as H  # noqa: E402
import tools_pop  # noqa: E402
from mc3_amd import engine  # noqa: E402


def measure(prob, n, warm):
    g = engine.DeviceChains(**prob)
    Z0, lp0 = tools_pop.population(g)
    g.set_population(Z0, lp0)
    g.run(warm)
    g.run(n)
    us = g.last_run_ms()*1e3/(n*g.thinning)
    pl = g.plan
    g.close()
    return us, pl


def sweep(name, shapes, build):
    rows = []
    for ndata, nchains in shapes:
        per = 1.2 if ndata <= 2048 else 2.0 + ndata/3000.0          # (rough us per iteration)
        n = int(max(200, min(6000, 40000/per)))
        prob = build(ndata, nchains, n + n//5 + 8)
        try:
            us, pl = measure(prob, n, n//5)
        except Exception as e:      # noqa: BLE001
            rows.append({'sweep': name, 'ndata': ndata, 'nchains': nchains, 'error': str(e)[:200]})
            continue
        rows.append({'sweep': name, 'ndata': ndata, 'nchains': nchains,
                     'sampler': prob['sampler'], 'us_per_iteration': round(us, 3),
                     'million_chain_steps_per_s': round(nchains/us, 2),
                     'giga_points_per_s': round(nchains*ndata/us/1e3, 2),
                     'plan': {k: pl[k] for k in ('fast', 'persistent', 'resident', 'tile',
                                                 'waves_per_chain', 'chains_per_workgroup',
                                                 'jit')}})
    # a cliff: below half the per-point rate of both neighbours (same number of chains)
    by_chains = {}
    for r in rows:
        if 'error' not in r:
            by_chains.setdefault(r['nchains'], []).append(r)
    for group in by_chains.values():
        group.sort(key=lambda r: r['ndata'])
        for k, r in enumerate(group):
            nb = [group[q]['giga_points_per_s'] for q in (k - 1, k + 1) if 0 <= q < len(group)]
            r['cliff'] = len(nb) == 2 and all(r['giga_points_per_s'] < 0.5*v for v in nb)
    for r in rows:
        print(json.dumps(r), flush=True)


def main():
    parts = sys.argv[1:] or ['gauss', 'wavelet', 'offlean']
    if 'gauss' in parts:
        nds = (250, 500, 1000, 1024, 1025, 1500, 2048, 2049, 3000, 4096, 6000, 8192, 12000, 16383,
               16384, 24000, 32768, 65536, 100000)
        sweep('gauss-demc', [(nd, 256) for nd in nds],
              lambda nd, nc, n: H.quad_problem(ndata=nd, nchains=nc, nsamples=nc*n, sampler='demc'))
        sweep('gauss-demc', [(nd, 1024) for nd in (1000, 2048, 4096, 8192, 16384, 32768)],
              lambda nd, nc, n: H.quad_problem(ndata=nd, nchains=nc, nsamples=nc*n, sampler='demc'))
        sweep('gauss-snooker', [(nd, 256) for nd in (1000, 2048, 4096, 8192, 16384, 32768)],
              lambda nd, nc, n: H.quad_problem(ndata=nd, nchains=nc, nsamples=nc*n,
                                               sampler='snooker'))
    if 'wavelet' in parts:
        sweep('wavelet-snooker', [(1 << m, 256) for m in range(6, 17)],
              lambda nd, nc, n: H.wavelet_problem(ndata=nd, nchains=nc, nsamples=nc*n))
        sweep('wavelet-demc', [(1 << m, 256) for m in (8, 10, 12, 13, 14)],
              lambda nd, nc, n: H.wavelet_problem(ndata=nd, nchains=nc, nsamples=nc*n,
                                                  sampler='demc'))
    if 'wlone' in parts:
        # the one-wave wavelet chains written out by hand (wavelet_one_kernel /
        # wavelet_one_other_kernel) against the general chain behind MC3HIP_NO_WL_ONE
        for smp in ('snooker', 'demc', 'mrw'):
            for off in (False, True):
                if off:
                    os.environ['MC3HIP_NO_WL_ONE'] = '1'
                sweep(f'wavelet-{smp}' + ('-general-chain' if off else ''),
                      [(1 << m, 256) for m in (6, 8, 10, 11, 12)],
                      lambda nd, nc, n, smp=smp: H.wavelet_problem(ndata=nd, nchains=nc,
                                                                   nsamples=nc*n, sampler=smp))
                os.environ.pop('MC3HIP_NO_WL_ONE', None)
    if 'offlean' in parts:
        # the shapes no hand-laid-out loop serves
        def poly(k, sampler, **kw):
            def f(nd, nc, n):
                prob = H.quad_problem(ndata=nd, nchains=nc, nsamples=nc*n, sampler=sampler, **kw)
                pad = k - 3
                prob.update(indep=prob['indep']/10.0,
                            params=np.concatenate([[10.0, -20.0, 10.0], np.zeros(pad)]),
                            pstep=np.concatenate([[1.0, 5.0, 10.0], np.full(pad, 1.0)]),
                            pmin=np.concatenate([[-10.0, -200.0, -1000.0], np.full(pad, -1e4)]),
                            pmax=np.concatenate([[40.0, 200.0, 1000.0], np.full(pad, 1e4)]))
                return prob
            return f
        sweep('offlean-mrw-6-free', [(1000, 256)], poly(6, 'mrw'))
        sweep('offlean-snooker-eps-6-free', [(1000, 256)], poly(6, 'snooker', fepsilon=0.01))
        sweep('offlean-demc-6-free', [(1000, 256)], poly(6, 'demc'))

        def fixed(nd, nc, n):
            prob = H.quad_problem(ndata=nd, nchains=nc, nsamples=nc*n, sampler='demc')
            prob['pstep'] = np.array([1.0, 0.5, 0.0])
            return prob
        sweep('offlean-fixed-coefficient', [(1000, 256), (2048, 256), (4096, 256)], fixed)


if __name__ == '__main__':
    main()
