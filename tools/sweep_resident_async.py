#!/usr/bin/env python3
"""Development aid: the one-launch tiled plan in step and with nobody in step across sizes --
where does which win?   python tools/sweep_resident_async.py > out.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import helpers as H  # noqa: E402
import tools_pop  # noqa: E402
from mc3_amd import engine  # noqa: E402

for k in (3, 10):
    for nchains in (128, 256, 512, 1024):
        for nd in (12000, 16384, 24000, 32768, 50000, 100000):
            for smp in ('demc', 'snooker'):
                row = dict(K=k, nchains=nchains, ndata=nd, sampler=smp)
                # (the async leg forced -- MC3HIP_RS_ASYNC=1 -- whatever decide_persist's rule
                # says; launches long enough that their beginning, in step, weighs little)
                for tag, env in (('async', None), ('in_step', '1')):
                    if env:
                        os.environ['MC3HIP_NO_RS_ASYNC'] = env
                        os.environ.pop('MC3HIP_RS_ASYNC', None)
                    else:
                        os.environ.pop('MC3HIP_NO_RS_ASYNC', None)
                        os.environ['MC3HIP_RS_ASYNC'] = '1'
                    n = int(os.environ.get('SWEEP_ITERS', 2000))
                    prob = H.linear_problem(ndata=nd, nmodel=k, nchains=nchains,
                                            nsamples=nchains*(n + n//5 + 8), sampler=smp)
                    g = engine.DeviceChains(**prob)
                    Z0, lp0 = tools_pop.population(g)
                    g.set_population(Z0, lp0)
                    g.run(n//5)
                    g.run(n)
                    row[tag] = round(g.last_run_ms()*1e3/n, 2)
                    row[tag + '_owner_waves'] = g.plan['owner_waves']
                    row['resident'] = g.plan['resident']
                    g.close()
                os.environ.pop('MC3HIP_NO_RS_ASYNC', None)
                os.environ.pop('MC3HIP_RS_ASYNC', None)
                print(json.dumps(row), flush=True)
