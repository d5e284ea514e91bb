#!/usr/bin/env python3
"""One-off burn-in: many more random shapes of the resident tiled plan than the committed
test carries (tests/test_gpu_resident.py::test_resident_random_problems), same checks.

    python tools/burnin_resident.py [first case] [number of cases]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_resident as T  # noqa: E402
from mc3_amd import engine  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for case in range(first, first + count):
    for name in ('test_resident_random_problems', 'test_resident_polynomial_random_problems'):
        try:
            getattr(T, name)(engine, case)
            print(case, name, 'ok', flush=True)
        except AssertionError as e:
            bad += 1
            print(case, name, 'FAILED', str(e)[:300], flush=True)
print('failures:', bad)
sys.exit(1 if bad else 0)
