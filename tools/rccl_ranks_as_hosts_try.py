#!/usr/bin/env python3
"""Development aid: does RCCL connect TWO ranks on ONE GPU when each is told it is a host of
its own (NCCL_HOSTID; tests/c3_worker.py MC3_TEST_RCCL_RANKS_AS_HOSTS)?  The ranks' output goes
to files that are printed whatever happens (a hang is cut off after `limit` seconds).
    python tools/rccl_ranks_as_hosts_try.py [nranks = 2] [limit = 90]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers as H  # noqa: E402
from mc3_amd import parallel  # noqa: E402

nranks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 90.0
nref, per_rank = 30, 32
prob = H.lean_shape_problem(per_rank*nranks, nref)
orc = H.make_oracle(prob)
Z0, lp0 = H.initial_population(orc)
orc.set_population(Z0, lp0)
orc.run(nthreads=8)
tmp = tempfile.mkdtemp()
out = os.path.join(tmp, 'c3')
np.savez(out + '.pop.npz', Z0=Z0, lp0=lp0)
base = dict(os.environ, OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0',
            MC3_TEST_RCCL_RANKS_AS_HOSTS='1', NCCL_DEBUG=os.environ.get('NCCL_DEBUG', 'WARN'))
rdv = parallel.new_rendezvous()
logs = [open(os.path.join(tmp, f'rank{r}.log'), 'w') for r in range(nranks)]
procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'c3_worker.py'), out,
                           str(nref), 'zlag=0', f'per_rank={per_rank}', 'exchange=rccl'],
                          env=parallel.rank_env(r, nranks, rdv, base=base),
                          stdout=logs[r], stderr=subprocess.STDOUT) for r in range(nranks)]
t0 = time.time()
while time.time() - t0 < limit and any(p.poll() is None for p in procs):
    time.sleep(0.5)
hung = [r for r, p in enumerate(procs) if p.poll() is None]
for p in procs:
    if p.poll() is None:
        p.kill()
for r in range(nranks):
    logs[r].close()
    text = open(os.path.join(tmp, f'rank{r}.log')).read()
    print(f'---- rank {r} (exit {procs[r].returncode}{", HUNG" if r in hung else ""})')
    print('\n'.join(l for l in text.splitlines() if 'NCCL INFO' not in l)[-3000:])
if all(os.path.exists(f'{out}.rank{r}.npz') for r in range(nranks)):
    for r in range(nranks):
        got = np.load(f'{out}.rank{r}.npz')
        assert np.array_equal(got['Z'], orc.Z), f'rank {r}: rows differ'
    print(f'RCCL all-gather over {nranks} ranks on one GPU (each a host of its own): '
          'replicas == oracle')
