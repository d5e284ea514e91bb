#!/usr/bin/env python3
"""Development aid: the generators of tests/test_gpu_fuzz.py on seeds the test suite does not
use, each problem compared with the oracle exactly (rows, counters, best fits).
    python tools/fuzz_hunt.py [first_seed] [count]        # prints one line per failure"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mc3_amd import engine  # noqa: E402
import helpers as H  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
async_plans = 0                 # (contexts that ran resident_async_kernel)
kernels = {}                    # (the chain kernel each context launched last)
for seed in range(first, first + count):
    kind = seed % 4 if 'big' not in sys.argv else 4
    if os.environ.get('FUZZ_KIND'):       # (one generator only, e.g. 2 = the wavelet likelihood)
        kind = int(os.environ['FUZZ_KIND'])
    if kind == 3:
        # round 4's kernels: the declared cadence, the support distribution, the run-time
        # specialised shapes
        prob = F.round4_problem(seed)[0]
    elif kind == 4:
        # ensembles beyond what is co-resident: two to five chains per wave (kernels_rows.h, the
        # turn-after-turn loops for two tiles), the shapes the lean kernels serve
        # (python tools/fuzz_hunt.py <first> <count> big)
        prob, _ = F.lean_problem(seed)
        rs = np.random.RandomState(seed)
        nch = int(rs.randint(2049, 10000))
        prob.update(nchains=nch, nsamples=nch*int(rs.randint(6, 16))*prob['thinning'],
                    zlag=int(rs.choice([0, 0, 1, 2, 5])))      # (strict, or the declared cadence)
    elif kind == 5:
        # round 5: 1 025 ... 16 384 points in the registers of several waves (FUZZ_KIND=5)
        prob = F.wide_problem(seed)
    elif kind == 6:
        # round 5: the resident tiled plan with nobody in step (FUZZ_KIND=6), whatever the size
        os.environ['MC3HIP_RS_ASYNC'] = '1'
        prob = F.resident_async_problem(seed)
    elif kind == 7:
        # round 5: the wide loops specialised at run time (FUZZ_KIND=7)
        prob = F.wide_offlean_problem(seed)
    elif kind == 8:
        # round 5: the models specialised at run time, beyond one register tile (FUZZ_KIND=8)
        prob = F.wide_models_problem(seed)
    elif kind == 9:
        # round 6: the headline loop beyond 1 024 chains (lean_demc_kernel BIG: the wait laid out
        # by hand) -- the plain DEMC shapes of the lean generator with 1 025 ... 2 048 chains
        # (FUZZ_KIND=9); tight bounds come with the generator
        for sub in range(1000):
            prob, _ = F.lean_problem(seed*1000 + sub)
            if prob['sampler'] == 'demc' and prob.get('thinning', 1) == 1 and \
                    len(prob['data']) <= 1024 and np.all(np.asarray(prob['pstep']) > 0) and \
                    not np.any(np.asarray(prob.get('priorlow', [0])) > 0):
                break
        rs = np.random.RandomState(seed)
        nch = int(rs.randint(1025, 2049))
        prob.update(nchains=nch, nsamples=nch*int(rs.randint(6, 90)))
    elif kind == 0:
        prob = F.random_problem(seed)
    elif kind == 1:
        prob, _ = F.lean_problem(seed)
    else:
        prob = F.random_problem(seed, wavelet_any_model=True)
        if seed % 9 == 2 and len(prob['data']) > 2:      # (a padded series: mc3hip_problem.wl_nreal)
            n = len(prob['data'])
            prob['wl_nreal'] = int(np.random.RandomState(seed).randint(n//2 + 1, n))
    if os.environ.get('FUZZ_TRACE'):      # (a fault kills the process: say which seed is next)
        print(f'seed {seed} kind {kind}', file=sys.stderr, flush=True)
    try:
        gpu = engine.DeviceChains(**prob)
        async_plans += 1 if gpu.plan.get('owner_waves', 0) > 0 else 0
        orc = H.make_oracle(prob, plan=gpu.plan)
        try:
            Z0, lp0 = H.initial_population(orc, seed=seed)
        except ValueError:
            gpu.close()
            continue
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        gpu.run(max(1, gpu.nzchain//3))
        gpu.run()
        F.compare_final(gpu, orc, f'seed {seed}')
        kernels[gpu.last_kernel] = kernels.get(gpu.last_kernel, 0) + 1
        gpu.close()
    except Exception as e:      # noqa: BLE001
        bad += 1
        print(f'FAIL seed {seed} kind {kind}: {str(e)[:300]}', flush=True)
print(f'{count} seeds from {first}: {bad} failures'
      + (f' ({async_plans} contexts with owner waves)' if async_plans else ''))
if os.environ.get('FUZZ_KERNELS'):
    for name, n in sorted(kernels.items(), key=lambda kv: -kv[1]):
        print(f'{n:6d}  {name}')
