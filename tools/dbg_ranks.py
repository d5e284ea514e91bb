#!/usr/bin/env python3
"""Development aid: tests/test_gpu_p2p.py lean_ranks_case at chosen sizes.
    python tools/dbg_ranks.py nranks per_rank zlag nref [ENV=VALUE ...]"""
import os
import sys
import tempfile
import time
import pathlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_p2p as T  # noqa: E402

nranks, per_rank, zlag, nref = (int(v) for v in sys.argv[1:5])
for kv in sys.argv[5:]:
    k, v = kv.split('=')
    os.environ[k] = v
t0 = time.time()
try:
    with tempfile.TemporaryDirectory() as d:
        T.lean_ranks_case(pathlib.Path(d), nranks, nref, zlag, per_rank=per_rank)
    print(f'OK   ranks={nranks} per_rank={per_rank} zlag={zlag} nref={nref} {sys.argv[5:]} '
          f'{time.time() - t0:.1f}s', flush=True)
except AssertionError as e:
    msg = str(e).strip().splitlines()
    if os.environ.get('MC3HIP_VERBOSE'):
        print('\n'.join(m[:300] for m in msg[:40]))
    print(f'FAIL ranks={nranks} per_rank={per_rank} zlag={zlag} nref={nref} {sys.argv[5:]} '
          f'{time.time() - t0:.1f}s :: {msg[-1][:200] if msg else ""}', flush=True)
