#!/usr/bin/env python3
"""Development aid: randomly drawn problems SHARDED over 2..4 ranks on the one GPU (peer-to-peer
hand-off where the shape has one), strict or in the declared cadence, rows for other ranks held
back 0 / 1.5 / 3 us, several chains per wave forced now and then -- every replica compared with
the oracle's run of the whole ensemble in the same cadence, exactly.
    python tools/fuzz_ranks.py [first_seed] [count]        # prints one line per failure"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers as H  # noqa: E402
import ranks  # noqa: E402
import fuzz_problems  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
tally = {}
for seed in range(first, first + count):
    nranks, delay, turns = fuzz_problems.ranks_setup(seed)
    prob, nref, zlag = fuzz_problems.ranks_problem(seed, nranks)
    label = (f"seed {seed}: {nranks} ranks x {prob['nchains']//nranks} chains, {prob['sampler']} "
             f"poly{prob['nmodel']} ndata={len(prob['data'])} thin={prob['thinning']} zlag={zlag} "
             f"delay={delay} turns={turns} pstep={prob['pstep']}")
    try:
        orc = H.make_oracle(prob, zlag=zlag)
        try:
            Z0, lp0 = H.initial_population(orc, seed=seed)
        except ValueError:
            continue
        orc.set_population(Z0, lp0)
        orc.run(nthreads=8)
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, 'f')
            np.savez(out + '.pop.npz', Z0=Z0, lp0=lp0)
            env = {}
            if delay:
                env['MC3HIP_TEST_REMOTE_DELAY_NS'] = str(delay)
            if turns:
                env['MC3HIP_PERSIST_CAPACITY'] = str(-(-(prob['nchains']//nranks)//turns))
            ranks.launch_ranks(nranks, 'fuzz_worker.py', out, seed, env=env or None, timeout=600)
            per = prob['nchains']//nranks
            for r in range(nranks):
                got = np.load(f'{out}.rank{r}.npz')
                s = slice(per*r, per*(r + 1))
                np.testing.assert_array_equal(got['Z'], orc.Z, err_msg=f'rank {r} Z')
                np.testing.assert_array_equal(got['lp'], orc.log_post, err_msg=f'rank {r} lp')
                np.testing.assert_array_equal(got['numaccept'], orc.numaccept[s])
                np.testing.assert_array_equal(got['X'], orc.X[s])
                np.testing.assert_array_equal(got['chisq'], orc.chisq[s])
                np.testing.assert_array_equal(got['best_chisq'], orc.best_chisq[s])
                key = (str(got['mode']), int(got['fast']), int(got['cpw']))
                tally[key] = tally.get(key, 0) + (1 if r == 0 else 0)
    except Exception as e:      # noqa: BLE001
        bad += 1
        print(f'FAIL {label}: {str(e)[:400]}', flush=True)
print(f'{count} seeds from {first}: {bad} failures; (mode, plan.fast, chains per wave) of the runs: '
      f'{dict(sorted(tally.items()))}')
