#!/usr/bin/env python3
"""Measure the REFERENCE (mc3 v3.2.1 itself) on this container's CPUs: BASELINE.md
section 3(1).  Runs only where /root/reference exists (it is imported from there with
oracle/_ref on the path for its compiled C; nothing is copied).

    python tools/ref_baseline.py [--repeats 3] [--out profiles/r02_reference_cpu.json]

Rate = slope between two ``nsamples`` values (removes fork / initial population /
update_output), median over the repeats.  Every run is its own process under a hard
timeout with retry: demc / mrw with ncpu > 1 can deadlock at the end of a run
(mc3/chain.py:297-299 vs mc3/mcmc_driver.py:352).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, time, io, contextlib
sys.dont_write_bytecode = True
sys.path.insert(0, %(ref)r)
sys.path.insert(0, '/root/reference')
import numpy as np
import mc3
from mc3.mcmc_driver import mcmc
import mc3.utils as mu

def quad(p, x):
    return p[0] + p[1]*x + p[2]*x**2.0

config, sampler, ncpu, nsamples = %(config)r, %(sampler)r, %(ncpu)d, %(nsamples)d
np.random.seed(314)
if config == 'C2':
    x = np.linspace(0, 10, 1000)
    y = quad([3.0, -2.4, 0.5], x)
    uncert = np.sqrt(np.abs(y))
    data = y + np.random.normal(0, uncert)
    params = np.array([10.0, -2.0, 0.1]); pstep = np.array([1.0, 0.5, 0.1])
    pmin = np.array([-10.0, -20.0, -10.0]); pmax = np.array([40.0, 20.0, 10.0])
    wlike = False
else:                       # C4: wavelet likelihood, ndata = 2^14
    x = np.linspace(0, 10, 16384)
    y = quad([3.0, -2.4, 0.5], x)
    data = y + np.random.normal(0, 1.0, 16384) + 0.5*np.sin(2*np.pi*x)
    uncert = np.ones(16384)
    params = np.array([3.0, -2.4, 0.5, 1.0, 0.5, 1.0])
    pstep = np.array([0.3, 0.1, 0.02, 0.0, 0.1, 0.1])
    pmin = np.array([-10.0, -20.0, -10.0, 0.0, 0.01, 0.01])
    pmax = np.array([40.0, 20.0, 10.0, 2.0, 10.0, 10.0])
    wlike = True
n = len(params)
z = np.zeros(n)
log = mu.Log(None, verb=0)
names = mu.default_parnames(n)
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    out = mcmc(data, uncert, quad, params, [x], {}, pmin, pmax, pstep, z, z, z,
               256, ncpu, nsamples, sampler, wlike, None, False, 0.0, 0.5, 0, 1,
               1.0, 0.0, 10, 'normal', None, False, log, names, names)
dt = time.perf_counter() - t0
steps = len(out['posterior'])
print('RESULT', steps, dt)
'''


def one_run(config, sampler, ncpu, nsamples, timeout):
    code = CHILD % dict(ref=os.path.join(ROOT, 'oracle', '_ref'), config=config,
                        sampler=sampler, ncpu=ncpu, nsamples=nsamples)
    for attempt in range(4):
        try:
            res = subprocess.run([sys.executable, '-c', code], capture_output=True,
                                 text=True, timeout=timeout, cwd='/tmp')
        except subprocess.TimeoutExpired:
            print(f'  (hung: {config} {sampler} ncpu={ncpu} nsamples={nsamples}, retry)',
                  flush=True)
            continue
        for line in res.stdout.splitlines():
            if line.startswith('RESULT'):
                _, steps, dt = line.split()
                return int(steps), float(dt), attempt
        raise RuntimeError(res.stderr[-2000:])
    raise RuntimeError('four hangs in a row')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--repeats', type=int, default=3)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r02_reference_cpu.json'))
    ap.add_argument('--quick', action='store_true')
    args = ap.parse_args()
    if not os.path.isdir('/root/reference/mc3'):
        sys.exit('the reference is not present on this box')
    ncpus = os.cpu_count()
    plans = [('C2', 'demc', 1, 100000, 300000), ('C2', 'demc', 7, 200000, 600000),
             ('C2', 'snooker', 1, 50000, 150000), ('C2', 'snooker', 7, 200000, 600000),
             ('C4', 'snooker', 1, 12800, 38400), ('C4', 'snooker', 7, 51200, 153600)]
    if args.quick:
        plans = plans[:2]
    rows = []
    for config, sampler, ncpu, n1, n2 in plans:
        slopes, whole, hangs = [], [], 0
        for rep in range(args.repeats):
            s1, t1, h1 = one_run(config, sampler, ncpu, n1, 900)
            s2, t2, h2 = one_run(config, sampler, ncpu, n2, 1800)
            slopes.append((s2 - s1)/(t2 - t1)/1e6)
            whole.append(s2/t2/1e6)
            hangs += h1 + h2
            print(f'{config} {sampler} ncpu={ncpu} rep {rep}: {s1} steps {t1:.2f} s, '
                  f'{s2} steps {t2:.2f} s -> slope {slopes[-1]:.4f} M/s '
                  f'(whole run {whole[-1]:.4f})', flush=True)
        slopes.sort()
        whole.sort()
        rows.append({'config': config, 'sampler': sampler, 'ncpu': ncpu,
                     'nsamples': [n1, n2], 'repeats': args.repeats,
                     'slope_Mcs_per_s_median': slopes[len(slopes)//2],
                     'slope_Mcs_per_s_all': slopes,
                     'whole_run_Mcs_per_s_median': whole[len(whole)//2],
                     'hung_runs_retried': hangs})
    rec = {'what': 'mc3 v3.2.1 (the reference itself), mcmc() wall clock, 256 chains',
           'box': f'build container, {ncpus} vCPU', 'time': time.ctime(), 'rows': rows}
    with open(args.out, 'w') as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main()
