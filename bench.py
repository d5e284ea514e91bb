#!/usr/bin/env python3
"""bench.py -- million chain-steps/sec of the MI355X chain-update path.

Workload (BASELINE.json configs[1], "C2"): quadratic 3-parameter fit,
ndata=1000, 256 chains per GPU, sampler='demc', thinning=1, fepsilon=0,
synthetic data as in examples/tutorial.py:22-46 of the reference (SURVEY.md
section 8d).  A "step" is one batch of `--refreshes` Z refreshes (default: 18 launches of
the persistent kernel, each 16 384 refreshes = 4.19 M chain-steps per GPU): in every refresh
every chain of the ensemble advances `thinning` iterations (proposal -> model -> chisq ->
Metropolis -> Z append).  Whatever --steps/--warmup are given, a step is long against launch
and synchronisation overheads, and the default run keeps the GPU busy for seconds on end.
Z (128 B per chain and refresh) is sized to a memory budget; a run longer than that recycles
the buffer INSIDE the timed region (mc3hip_recycle: the host has no use for the rows of a
benchmark; the last refreshes move to the front, the rest is unwritten again, the chains
carry on) -- config.recycles says how often.

    python bench.py [--gpus N --steps K --warmup W] [--workload c2|c3|c5]

N > 1, either way of starting it works:
  * ``python bench.py --gpus N`` as ONE plain process: it never touches the GPU itself,
    starts N rank processes (RANK / WORLD_SIZE / MC3HIP_RDV_FILE / _SECRET in their environment),
    relays rank 0's JSON line, ends everything on a watchdog and exits non-zero if any
    rank failed;
  * ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N``:
    the processes the launcher started are the ranks.
The ranks meet through mc3_amd.parallel.SocketComm (plain TCP; no torch in this file).

N > 1, what is measured (c2 = weak scaling, 256*N chains; c3 / c5 = strong scaling, BASELINE
configs 3 and 5: 2048 chains x ndata 1000 and 512 chains x 1e5 points x 10 coefficients
split over the ranks).  Rank r owns a contiguous block of chains and a full replica of Z:
  * `value`: the peer-to-peer hand-off (the persistent kernel stores every new Z row into
    every rank's replica over xGMI; no collective), falling back to RCCL by consensus;
  * `config.collective_rccl`: a second timed segment through ONE in-place RCCL all-gather of
    the new Z rows per refresh, replayed from a hipGraph -- the north star's collective --
    with the rank count RCCL itself reports;
  * `config.single_gpu_same_ensemble`: the whole ensemble in ONE context on rank 0's GPU,
    the honest yardstick for "is N GPUs faster than one".

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHAINS_PER_GPU = 256
# the metric exactly as BASELINE.json names it
METRIC = 'million chain-steps/sec (nchains×iters/s), 256-chain quad fit ndata=1000'
NDATA = 1000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # MI355X_MICROARCH.md: dense FP64 vector / matrix
LAUNCH_REFRESHES = 16384       # refreshes one launch of a persistent kernel covers
STEP_LAUNCHES = 18             # persistent launches per step (C2: ~0.35 s of GPU time)
PMC_FILES = ('r06_c2_pmc.json', 'r05_c2_pmc.json', 'r04_c2_pmc.json', 'r03_c2_pmc.json', 'r02_c2_pmc.json',
             'r01g_c2_pmc.json')
SIMDS = 1024                   # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4                # MI355X_MICROARCH.md: peak engine clock (the FP64 issue peak:
                               # one 64-lane FP64 instruction per SIMD every 4 cycles)


def quad(p, x):
    return p[0] + p[1]*x + p[2]*x**2.0


def c2_problem(nchains, nrefresh, thinning=1):
    rs = np.random.RandomState(314)
    x = np.linspace(0, 10, NDATA)
    y = quad([3.0, -2.4, 0.5], x)
    uncert = np.sqrt(np.abs(y))
    data = y + rs.normal(0, uncert)
    return dict(data=data, uncert=uncert, indep=x,
                params=np.array([10.0, -2.0, 0.1]),
                pstep=np.array([1.0, 0.5, 0.1]),
                pmin=np.array([-10.0, -20.0, -10.0]),
                pmax=np.array([40.0, 20.0, 10.0]),
                nchains=nchains, nsamples=nchains*nrefresh*thinning,
                thinning=thinning, sampler='demc', model='poly', seed=12345,
                hsize=10)


def c5_problem(nchains, nrefresh, thinning=1, ndata=100000, nmodel=10):
    """BASELINE config 5: linear model on a Legendre / cosine design
    matrix, unit-variance noise, Gaussian priors (sigma = 1) on three parameters, bounds +-5."""
    rs = np.random.RandomState(7)
    t = np.linspace(-1, 1, ndata)
    A = np.array([np.polynomial.legendre.legval(t, [0]*(k//2) + [1]) if k % 2 == 0
                  else np.cos(np.pi*(k//2 + 1)*t) for k in range(nmodel)])
    p_true = rs.uniform(-1, 1, nmodel)
    signal = np.zeros(ndata)
    for k in range(nmodel):           # (term by term: every rank must build the same bits)
        signal = signal + p_true[k]*A[k]
    data = signal + rs.normal(0, 1, ndata)
    prior, plow, pup = np.zeros(nmodel), np.zeros(nmodel), np.zeros(nmodel)
    prior[:3], plow[:3], pup[:3] = p_true[:3], 1.0, 1.0
    return dict(data=data, uncert=np.ones(ndata), indep=A,
                params=p_true + rs.normal(0, 0.05, nmodel), pstep=np.full(nmodel, 0.05),
                pmin=np.full(nmodel, -5.0), pmax=np.full(nmodel, 5.0),
                prior=prior, priorlow=plow, priorup=pup,
                nchains=nchains, nsamples=nchains*nrefresh*thinning, thinning=thinning,
                sampler='demc', model='linear', seed=2024, hsize=10)


def c4_problem(nchains, nrefresh, thinning=1, ndata=16384):
    """BASELINE config 4: the quadratic on linspace(0, 10, 2^14), white +
    red noise, (gamma, sigma_r, sigma_w) appended with gamma fixed, wavelet likelihood,
    snooker (tests/helpers.py wavelet_problem: the shape the parity tests run)."""
    rs = np.random.RandomState(99)
    x = np.linspace(0, 10, ndata)
    data = quad([3.0, -2.4, 0.5], x) + rs.normal(0, 1.0, ndata) + 0.5*np.sin(2*np.pi*x)
    return dict(data=data, uncert=np.ones(ndata), indep=x,
                params=np.array([3.0, -2.4, 0.5, 1.0, 0.5, 1.0]),
                pstep=np.array([0.3, 0.1, 0.02, 0.0, 0.1, 0.1]),
                pmin=np.array([-10.0, -20.0, -10.0, 0.0, 0.01, 0.01]),
                pmax=np.array([40.0, 20.0, 10.0, 2.0, 10.0, 10.0]),
                nchains=nchains, nsamples=nchains*nrefresh*thinning, thinning=thinning,
                sampler='snooker', wlike=True, model='poly', seed=4321, hsize=10)


# name -> (problem builder, chains: per GPU (weak) or in total (strong), scaling, algorithmic
# bytes per chain-step by SURVEY.md section 8d: 8*S*ndata + 8*nfree*(R+1) + 8, description)
WORKLOADS = {
    'c2': (c2_problem, CHAINS_PER_GPU, 'weak', 8*3*NDATA + 8*3*(3 + 1) + 8,
           "C2: quadratic 3-param fit, ndata=1000, 256 chains/GPU, sampler='demc', fepsilon=0"),
    'c3': (c2_problem, 2048, 'strong', 8*3*NDATA + 8*3*(3 + 1) + 8,
           "C3: quadratic 3-param fit, ndata=1000, 2048 chains split over the GPUs, "
           "sampler='demc', fepsilon=0"),
    # (not a BASELINE configuration: a large ensemble per GPU, several chains per wave --
    # kernels_rows.h -- sharded peer to peer)
    'c3big': (c2_problem, 4096, 'weak', 8*3*NDATA + 8*3*(3 + 1) + 8,
              "quadratic 3-param fit, ndata=1000, 4096 chains/GPU, sampler='demc', fepsilon=0"),
    'c5': (c5_problem, 512, 'strong', 8*12*100000 + 8*10*(3 + 1) + 8,
           "C5: 10-param linear model, ndata=1e5, 512 chains split over the GPUs, Gaussian "
           "priors + bounds, sampler='demc'"),
}
BYTES_PER_CHAIN_STEP = WORKLOADS['c2'][3]


def initial_population(prob, eval_chisq, expand=None):
    """M0 = hsize*nchains in-bounds draws around the initial guess
    (mcmc_driver.py:229-262), chisq from the device."""
    rs = np.random.RandomState(5)
    M0 = prob['hsize']*prob['nchains']
    x0, sig = prob['params'], prob['pstep']
    nf = len(x0)
    Z0 = np.empty((0, nf))
    while len(Z0) < M0:
        trial = rs.normal(x0, sig, (2*M0, nf))
        ok = np.all((trial >= prob['pmin']) & (trial <= prob['pmax']), axis=1)
        Z0 = np.concatenate([Z0, trial[ok]])
    Z0 = np.ascontiguousarray(Z0[:M0])
    return Z0, -0.5*eval_chisq(Z0)


def population(g, seed=5):
    """M0 in-bounds draws of the FREE parameters around the start with finite chisq, evaluated
    on the device (mcmc_driver.py:229-262)."""
    rs = np.random.RandomState(seed)
    ifree = g.ifree
    Z0, lp0 = np.empty((0, g.nfree)), np.empty(0)
    while len(Z0) < g.M0:
        trial = rs.normal(g.params[ifree], g.pstep[ifree], (2*g.M0, g.nfree))
        full = g.expand_free(trial)
        ok = np.all((full >= g.pmin) & (full <= g.pmax), axis=1)
        lp = -0.5*g.eval_chisq(full[ok])
        fin = np.isfinite(lp)
        Z0 = np.concatenate([Z0, trial[ok][fin]])
        lp0 = np.concatenate([lp0, lp[fin]])
    return np.ascontiguousarray(Z0[:g.M0]), np.ascontiguousarray(lp0[:g.M0])


def _usable_cpus():
    return len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') \
        else (os.cpu_count() or 1)


def _cpu_quota_cores():
    """CPUs' worth of time the cgroup grants per period (None = no quota): a box may SHOW 256
    CPUs and GRANT 16 -- threads beyond the quota only take turns (and get the whole process
    frozen for the rest of the period: profiles/r06_e2e_stall.txt)."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        return None if quota == 'max' else int(quota)/int(period)
    except (OSError, ValueError):
        return None


def cpu_baseline(target_seconds=8.0):
    """The oracle (oracle/oracle_lockstep.c: the same lockstep chain update in plain C) on this
    box's host cores, the same C2 workload, bounded to ~target_seconds per figure -- TWICE:

    * ``layout: processes`` (the reference's layout): worker PROCESSES with the
      chains strided over them, worker i advancing chains i, i + ncpu, ... one after another
      (mc3/chain.py:163,190-191) and all of them meeting once per refresh (the reference:
      once per iteration through pipes to the hub, mcmc_driver.py:300-307; here a barrier in
      shared memory, the cheapest meeting there is) -- ls_run_worker;
    * ``layout: threads``: OpenMP threads of one process over the chains (ls_run).

    ``value`` is the faster of the two (``layout`` says which; ``kind`` is "port" for both: the
    oracle, not a build of the reference), the other is reported beside it
    (``port_threads`` / ``port_processes``).  Each with the worker / thread count that is
    fastest in a short calibration over 1..usable cores (a container may see more CPUs than it may use).  Runs in a child process of its own
    that never loads the HIP runtime (bench.py --cpu-baseline-child): the workers are forked."""
    from oracle import lockstep as ls
    usable = _usable_cpus()
    nmax = 60000
    prob = c2_problem(CHAINS_PER_GPU, 2*nmax + 4096)

    def fresh(shared):
        orc = ls.LockstepRun(shared=shared, **prob)
        Z0, lp0 = initial_population(prob, lambda Z: orc.eval_chisq(Z))
        orc.set_population(Z0, lp0)
        return orc

    def measure(step, label):
        """step(count, n) -> seconds for n refreshes with `count` workers."""
        step(1, 20)
        best, best_rate = 1, 100/step(1, 100)
        for k in (2, 4, 8, 16, 32, 64, 128, 256):
            if k > min(usable, CHAINS_PER_GPU):
                break
            r = 200/step(k, 200)
            if r > best_rate:
                best, best_rate = k, r
            elif r < 0.5*best_rate:
                break
        chunk = int(max(200, best_rate))              # about one second of work
        n, dt = 0, 0.0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < target_seconds and n + chunk <= nmax:
            dt += step(best, chunk)
            n += chunk
        # (`kind`: the contract's word -- the oracle, not a build of the reference; `layout`: which
        # of the two ways of putting it on the cores)
        return {'value': CHAINS_PER_GPU*n/dt/1e6, 'unit': 'million chain-steps/sec',
                'cores': best, 'kind': 'port', 'layout': label, 'iterations': n, 'seconds': dt}

    orc_p = fresh(True)
    procs = measure(lambda k, n: orc_p.run_processes(n, k), 'processes')
    orc_t = fresh(False)

    def threads(k, n):
        t0 = time.perf_counter()
        orc_t.run(nrefresh=n, nthreads=k)
        return time.perf_counter() - t0
    thr = measure(threads, 'threads')
    procs['sample'] = (
        f'{procs.pop("iterations")} iterations of the same 256-chain C2 workload '
        f'({procs.pop("seconds"):.1f} s), oracle_lockstep.c ls_run_worker: {procs["cores"]} worker '
        'PROCESSES, chains strided i + j*ncpu over them (mc3/chain.py:163,190-191), one '
        f'meeting per refresh in shared memory (best of a calibration up to {usable} usable '
        'CPUs).  mc3\'s own multiprocessing path cannot travel to the GPU box and is measured '
        'in the build container: BASELINE.md section 2b')
    thr['sample'] = (
        f'{thr.pop("iterations")} iterations ({thr.pop("seconds"):.1f} s), the same C loop under '
        f'{thr["cores"]} OpenMP THREADS of one process, a barrier per refresh')
    host = {'cpus_visible': usable, 'cpu_quota_cores': _cpu_quota_cores()}
    procs['host'] = thr['host'] = host
    # The figure of record is the FASTER of the two -- a GPU / CPU ratio taken from the slower
    # one would flatter the GPU (round 5: 1.34 by processes beside 1.72 by threads in one run);
    # the other stays beside it under its own name.
    if thr['value'] > procs['value']:
        thr['port_processes'] = procs
        return thr
    procs['port_threads'] = thr
    return procs


E2E_CASES = ('c2_1e7_samples', 'c5_1e5_samples', 'c5_1e6_samples')


def end_to_end_child(name):
    """ONE end-to-end case in this (fresh) process: mc3_amd.sample() as a user calls it -- an
    ordinary Python callable as the model, the initial population, ten reports with the
    Gelman-Rubin test, the posterior statistics, the output dictionary -- after one small
    warm-up call (library load, first launches).  Prints one JSON line."""
    import io
    import contextlib
    import mc3_amd
    from mc3_amd import utils as mu
    if name.startswith('c2'):
        p2 = c2_problem(CHAINS_PER_GPU, 1)
        kw = dict(data=p2['data'], uncert=p2['uncert'], func=quad, params=p2['params'],
                  indparams=[p2['indep']], pstep=p2['pstep'], pmin=p2['pmin'], pmax=p2['pmax'],
                  sampler='demc', burnin=100, nchains=CHAINS_PER_GPU, grtest=True, seed=3)
        nsamples = 1e7
    else:
        p5 = c5_problem(512, 1)
        kw = dict(data=p5['data'], uncert=p5['uncert'], func=lambda p, A: p @ A,
                  params=p5['params'], indparams=[p5['indep']], pstep=p5['pstep'],
                  pmin=p5['pmin'], pmax=p5['pmax'], prior=p5['prior'],
                  priorlow=p5['priorlow'], priorup=p5['priorup'],
                  sampler='demc', burnin=20, nchains=512, grtest=True, seed=3)
        nsamples = 1e6 if '1e6' in name else 1e5
    with contextlib.redirect_stdout(io.StringIO()):
        mc3_amd.sample(nsamples=kw['nchains']*(kw['burnin'] + 100), log=mu.Log(verb=0), **kw)
        time.sleep(0.3)
        best = None
        for _ in range(3):                   # (the best of three: the box's other tenants)
            t0 = time.perf_counter()
            res = mc3_amd.sample(nsamples=nsamples, log=mu.Log(verb=0), **kw)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
    print(json.dumps({'wall_s': best, 'samples': int(len(res['posterior'])),
                      'million_samples_per_s': len(res['posterior'])/best/1e6,
                      'acceptance_rate': float(res['acceptance_rate']),
                      'host_threads': __import__('mc3_amd.sampler_driver', fromlist=['x'])
                      .host_threads()}), flush=True)


def end_to_end():
    """mc3_amd.sample() end to end -- wall seconds, NOT part of `value`: the rows next to the
    hot path measured where the driver can see them.  Each case in
    a child process of its own (round 6: in one process the second case inherited the first's
    freed multi-GB allocations and ran 2-3x slower)."""
    out = {}
    for name in E2E_CASES:
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), '--e2e-child', name],
                                 capture_output=True, text=True, timeout=300)
            out[name] = json.loads(res.stdout.strip().splitlines()[-1])
        except Exception as e:      # noqa: BLE001 -- an extra: never fails the bench line
            out[name] = {'error': str(e)[:200]}
    return out


# ---- the other single-GPU BASELINE configurations, each in a child process of its own ----------
# (only C2 was driver-timed.  A fresh process per configuration: after the
# multi-GB allocations of an earlier configuration have been freed, the streaming kernels of C5
# were seen to run 2-3x slower in the same process.)
def _fp64_per_chain_step(name):
    """FP64 wave-instructions (fma + mul + add) per chain-step of a configuration's kernel, from
    the newest committed PMC pass of it (rocprofv3 --pmc cannot run inside this process)."""
    for rnd in ('r06', 'r05', 'r04'):
        q = os.path.join(ROOT, 'profiles', f'{rnd}_{name}_pmc.json')
        if os.path.exists(q):
            with open(q) as f:
                rec = json.load(f)
            v = rec.get('derived', {}).get('f64_fma_mul_add_per_chain_step')
            if v:
                return float(v), 'profiles/' + os.path.basename(q), rec['kernel']['Kernel_Name']
    return None, None, None


CONFIGS = {
    # name: (problem builder, chains, warm-up refreshes, timed refreshes, algorithmic bytes per
    # chain-step, text)
    'c3one': (c2_problem, 2048, 16384, 65536, 8*3*NDATA + 8*3*(3 + 1) + 8,
              "C3's ensemble in ONE context: quadratic fit, ndata=1000, 2048 chains, demc, "
              "strict lockstep"),
    'c4': (c4_problem, 256, 4096, 32768, 8*2*16384 + 8*5*(3 + 1) + 8,
           "C4: wavelet likelihood (_dwt), ndata=2^14, 256 chains, snooker"),
    'c5': (c5_problem, 512, 1024, 8192, 8*12*100000 + 8*10*(3 + 1) + 8,
           "C5's ensemble in ONE context: 10-param linear model, ndata=1e5, 512 chains, "
           "Gaussian priors + bounds, demc"),
}


def config_child(name):
    """One configuration on GPU 0: warm-up launches, then `timed` refreshes as whole persistent
    launches; wall clock around them (stream drained on both sides) and HIP events on the
    library's stream (mc3hip_last_run_ms, summed over the launches).  Prints ONE JSON line."""
    from mc3_amd import engine
    build_problem, nchains, nwarm, ntimed, bytes_per_step, text = CONFIGS[name]
    prob = build_problem(nchains, nwarm + ntimed + 8)
    with engine.DeviceChains(**prob) as g:
        Z0, lp0 = population(g)
        g.set_population(Z0, lp0)
        g.run(nwarm)
        plan = dict(g.plan)
        per = plan['persistent'] or ntimed
        gpu_ms, launches, left = 0.0, 0, ntimed
        g.sync()
        t0 = time.perf_counter()
        while left:
            m = min(left, per)
            g.run(m, sync=False)
            gpu_ms += g.last_run_ms()           # (waits for that launch's end event)
            launches += 1
            left -= m
        g.sync()
        dt = time.perf_counter() - t0
        st = g.state()
        nd, nmodel, thinning = g.ndata, g.nmodel, g.thinning
        done = g.refresh_done
        launched = g.last_kernel        # (the library's own word: mc3hip_debug_last_kernel)
    steps = nchains*ntimed*thinning
    rate = steps/dt/1e6
    rate_ev = steps/(gpu_ms*1e-3)/1e6
    kernel = (('mc3hip::resident_async_kernel' if plan.get('owner_waves') else
               'mc3hip::resident_tile_kernel') if plan.get('resident') else
              {3: 'mc3hip::wavelet_pipe_kernel', 2: 'mc3hip::lean_demc_kernel',
               1: 'mc3hip::persist_fast_kernel', 0: 'mc3hip::persist_kernel',
               4: 'mc3hip::resident_tile_kernel'}.get(plan['fast'], 'mc3hip::persist_kernel'))
    out = {
        'config': name, 'workload': text, 'nchains': nchains, 'ndata': nd,
        'refreshes_timed': ntimed, 'launches': launches, 'kernel': kernel,
        'kernel_launched': launched,
        'million_chain_steps_per_s': rate, 'us_per_iteration': dt/ntimed/thinning*1e6,
        # HIP events on the library's stream over the same launches
        'avg_launch_us': gpu_ms*1e3/launches,
        'us_per_iteration_hip_events': gpu_ms*1e3/ntimed/thinning,
        'million_chain_steps_per_s_hip_events': rate_ev,
        'acceptance_rate': float(st['numaccept'].sum())/(nchains*done*thinning),
        'plan': {k: plan[k] for k in ('fast', 'persistent', 'resident', 'waves_per_chain',
                                      'chains_per_workgroup', 'tile', 'recoveries')},
        # the contractual figure (SURVEY.md 8d algorithmic bytes against the HBM peak) --
        # the operands are chain-invariant and sit on chip, so it may exceed 1: not a limiter
        'hbm_contract': {'bytes_per_chain_step': bytes_per_step,
                         'achieved_GBps': rate_ev*1e6*bytes_per_step/1e9,
                         'frac_of_peak': rate_ev*1e6*bytes_per_step/1e9/HBM_PEAK_GBS},
    }
    if name == 'c5':
        useful = rate_ev*1e6*2.0*nmodel*nd/1e12
        out['limiter'] = {'bound': 'fp64 matrix / vector pipe (the K = 10 contraction)',
                          'achieved_TFLOPs': useful, 'peak_TFLOPs': FP64_PEAK_TFLOPS,
                          'frac': useful/FP64_PEAK_TFLOPS}
    else:
        f64, src, kname = _fp64_per_chain_step(name)
        out['limiter'] = {
            'bound': 'fp64 vector issue (one 64-lane FP64 instruction per SIMD per 4 cycles, '
                     f'{SIMDS} SIMDs at {CLOCK_GHZ} GHz)',
            'fp64_instructions_per_chain_step': f64, 'instruction_count_from': src,
            'instruction_count_kernel': kname,
            'frac': None if f64 is None else rate_ev*1e6*f64*4/(SIMDS*CLOCK_GHZ*1e9)}
    print(json.dumps(out), flush=True)


def run_children(names, timeout=600):
    """bench.py --config-child NAME for every name, one after another; {name: record}."""
    out = {}
    for name in names:
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), '--config-child',
                                  name], capture_output=True, text=True, timeout=timeout)
            line = next((ln for ln in reversed(res.stdout.splitlines()) if ln.startswith('{')),
                        None)
            out[name] = json.loads(line) if res.returncode == 0 and line else \
                {'error': (res.stderr or res.stdout)[-300:] or f'exit code {res.returncode}'}
        except Exception as e:      # noqa: BLE001 -- an extra: never fails the bench line
            out[name] = {'error': str(e)[:300]}
    return out


def hbm_peak_probe():
    """tools/probes/hbm_read_probe (built by __graft_entry__.build): streaming-read bandwidth of
    THIS box, the measured denominator next to the 8 TB/s of the specification."""
    from mc3_amd import build
    exe = build.probe_path('hbm_read_probe')
    try:
        res = subprocess.run([exe, '4', '8'], capture_output=True, text=True, timeout=120)
        rec = json.loads(res.stdout.strip().splitlines()[-1])
        return {'GBps': rec['GBps_best'], 'GBps_mean': rec['GBps_mean'], 'bytes': rec['bytes'],
                'shape': rec['shape'], 'source': 'tools/probes/hbm_read_probe.hip, this run'}
    except Exception as e:      # noqa: BLE001
        return {'error': str(e)[:200]}


# ---- N > 1 started as one plain process: the parent of the ranks ----------------------------
def launch_ranks(args, argv):
    """Start one process per GPU, relay rank 0's JSON line, watch over them.  This process
    never initialises the GPU (and never replaces itself: the ranks are children)."""
    from mc3_amd import parallel
    n = args.gpus
    rdv = parallel.new_rendezvous()
    deadline = time.monotonic() + float(os.environ.get('MC3HIP_BENCH_WATCHDOG', 900)) + 60.0
    procs = []
    for r in range(n):
        env = parallel.rank_env(r, n, rdv)
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + argv, env=env,
            stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True,
            start_new_session=True))
    failed = []
    try:
        while not failed and any(p.poll() is None for p in procs):
            failed = [f'rank {r} exited with {p.returncode}' for r, p in enumerate(procs)
                      if p.poll() not in (None, 0)]
            if not failed and time.monotonic() > deadline:
                failed = ['watchdog: the ranks did not finish in time']
            # (rank 0's pipe holds one line: it cannot fill up and block the child)
            time.sleep(0.05)
        if failed:
            time.sleep(0.5)         # (ranks that die of the first one's death: say so too)
            failed = [f'rank {r} exited with {p.returncode}' for r, p in enumerate(procs)
                      if p.poll() not in (None, 0)] or failed
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()            # (exactly the processes started here)
        for p in procs:
            p.wait()
    out = procs[0].stdout.read() if procs[0].stdout else ''
    if not failed:
        failed = [f'rank {r} exited with {p.returncode}' for r, p in enumerate(procs)
                  if p.returncode != 0]
    failed = '; '.join(failed) or None
    line = next((ln for ln in reversed(out.splitlines()) if ln.startswith('{')), None)
    if failed or line is None:
        print(f'bench.py --gpus {n}: {failed or "rank 0 printed no result"}', file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


class DryEngine:
    """--dry-run: stands in for DeviceChains so that the launcher, the rendezvous and the
    reporting can be driven on a box without a GPU (tests/test_bench_launcher.py).  It
    computes nothing and its line says so (`data: dry-run`): never a measurement."""

    def __init__(self, nchains, nlocal, chain0, nrefresh, **_):
        self.nchains, self.nlocal, self.chain0, self.M0 = nchains, nlocal, chain0, 10*nchains
        self.refresh_done = 0
        self.plan = {'persistent': LAUNCH_REFRESHES, 'fast': 2, 'comm_graph': 0,
                     'graph_nodes': 0, 'waves_per_chain': 1, 'recoveries': 0, 'resident': 0}

    def run(self, n, sync=True):
        self.refresh_done += n
        time.sleep(1e-7*n)

    def sync(self):
        pass

    def last_run_ms(self):
        return 1.0

    def rows(self, row0, nrows):
        return np.ones((nrows, 3)), np.zeros(nrows), np.zeros(nrows, np.int32)

    def put_rows(self, row0, Z, log_post=None):
        pass

    def state(self):
        return {'numaccept': np.full(self.nlocal, self.refresh_done//4)}

    def close(self):
        pass


def run_rank(args):
    # stdout carries ONE JSON line: whatever native libraries print there meanwhile (RCCL's
    # banners) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    shared_device = 'MC3HIP_BENCH_DEVICE' in os.environ
    if shared_device:                            # testing: several ranks on one GPU
        local_rank = int(os.environ['MC3HIP_BENCH_DEVICE'])
    args.gpus = world

    from mc3_amd import parallel
    engine = None
    if not args.dry_run:
        from mc3_amd import engine

    # MC3HIP_BENCH_FORCE_COMM=1 drives the RCCL code path (communicator, one all-gather per
    # refresh) with a single rank, for testing
    force_comm = bool(os.environ.get('MC3HIP_BENCH_FORCE_COMM')) and world == 1
    ctl = parallel.SocketComm.from_env() if world > 1 else parallel.SingleRank()

    if world > 1:
        # a multi-rank run that stops making progress must not hang the node: dump every
        # thread's stack and exit after MC3HIP_BENCH_WATCHDOG seconds (default 900)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get('MC3HIP_BENCH_WATCHDOG', 900)),
                                          exit=True)

    build_problem, wl_chains, scaling, bytes_per_step, wl_text = WORKLOADS[args.workload]
    if scaling == 'weak':
        nlocal, nchains = wl_chains, wl_chains*world
    else:
        if wl_chains % world:
            sys.exit(f'--workload {args.workload}: {wl_chains} chains do not split over '
                     f'{world} ranks')
        nlocal, nchains = wl_chains//world, wl_chains
    chain0 = rank*nlocal
    nsteps, nwarm = max(1, args.steps), max(0, args.warmup)
    row_bytes = nchains*(16*8 + 8 + 4)            # padded Z row + log_post + zchain
    # Z is sized to a memory budget -- never more than half of the free memory of a GPU,
    # shared by the ranks on it -- and recycled when a step does not fit any more
    # (several ranks: every rank maps every peer's replica -- 32 GiB each keeps eight of them
    # quick to export, attach and check)
    budget = (96 << 30) if world == 1 else (32 << 30)
    if engine is not None:
        free, _ = engine.device_memory(local_rank)
        budget = min(budget, free//2//(world if shared_device else 1))
    if args.dry_run:
        if os.environ.get('MC3HIP_BENCH_DRY_FAIL_RANK') == str(rank):
            sys.exit(3)                           # (test hook: a rank that dies)
        R = args.refreshes or 4
    elif args.workload == 'c5':
        R = args.refreshes or 100                 # (~30 us per iteration: 100 per step)
    elif world == 1:
        R = args.refreshes or LAUNCH_REFRESHES*STEP_LAUNCHES
    else:
        # several ranks: the whole run fits Z (re-arming the rows of a 2048-chain ensemble
        # costs ~0.1 us per refresh against ~0.65 us of work: kept out of the timed region)
        fit = budget//(row_bytes*(nsteps + nwarm))
        R = args.refreshes or int(LAUNCH_REFRESHES*min(6, fit//LAUNCH_REFRESHES)
                                  if fit >= LAUNCH_REFRESHES else max(64, fit))
    nref_timed, nref_warm = nsteps*R, nwarm*R
    nrefresh = nref_timed + nref_warm
    fit = max(64, int(budget//row_bytes) - 64)
    capacity = min(nrefresh + 64, max(fit, 2*min(R, LAUNCH_REFRESHES) + 64))
    prob = build_problem(nchains, capacity, args.thinning)
    comm = None
    exchange = None

    ctl_opts = {}

    def make_engine(p2p, prob=prob, chain0=chain0, nlocal=nlocal):
        if args.dry_run:
            return DryEngine(prob['nchains'], nlocal, chain0, nrefresh)
        # (what the communicator asks of its engine: the declared cadence of a sharded run,
        # fine-grained Z, no courier wave when more than two ranks share a device)
        opts = dict(ctl_opts)
        if not p2p:
            opts.pop('p2p', None)
            opts.pop('no_courier', None)
        g = engine.DeviceChains(device=local_rank, chain0=chain0, nlocal=nlocal, **opts, **prob)
        Z0, lp0 = initial_population(prob, lambda Z: g.eval_chisq(g.expand_free(Z)))
        g.set_population(Z0, lp0)
        return g

    if world > 1:
        # preferred: the in-launch hand-off across GPUs (peer-to-peer stores into every
        # rank's Z replica); validated on a short run, else one RCCL all-gather per
        # refresh.  Every rank takes the same branch.
        comm = ctl
        comm.wants_p2p = not os.environ.get('MC3HIP_BENCH_NO_P2P') and not args.dry_run
        # (`value` is the STRICT lockstep unless --zlag says otherwise: the same random process
        # at every N, the library's default on any number of ranks; the declared cadence is
        # measured beside it -- cadence_segment -- and named `values.cadence`)
        ctl_opts.update(comm.engine_options(local_rank, args.thinning, args.zlag))
        gpu = make_engine(comm.wants_p2p)
        comm.setup(gpu)
        if comm.mode == 'p2p':
            try:
                gpu.run(64, sync=False)
                comm.quiesce(gpu)           # (recovers by itself if a launch gave up)
                ok = comm.mode == 'p2p'     # ... but then the run stays on that exchange
            except Exception as e:          # noqa: BLE001
                print(f'[rank {rank}] peer-to-peer hand-off failed ({e}); using RCCL',
                      file=sys.stderr, flush=True)
                ok = False
            if not comm._all_ok(ok) and comm.mode == 'p2p':
                gpu.close()
                comm.wants_p2p = False
                gpu = make_engine(False)
                comm.setup(gpu)
        exchange = comm.mode
    else:
        ctl_opts.update(ctl.engine_options(local_rank, args.thinning, args.zlag))
        gpu = make_engine(False)
        if force_comm:
            gpu.comm_init(engine.comm_unique_id(), 0, 1)
            exchange = 'rccl'

    def barrier(g=None):
        # multi-rank: quiesce() drains the library's stream, lets the ranks agree that
        # the segment completed everywhere (peer-to-peer stores: every peer has finished
        # too) and, if a launch gave up anywhere, rolls all of them back and redoes it
        g = g or gpu
        if comm is not None and g is gpu:
            comm.quiesce(g)
        g.sync()
        ctl.barrier()

    recycles = [0]
    done = [0 if args.dry_run else gpu.refresh_done]
    gpu_ms_acc = [0.0]
    run_sizes = []                 # refreshes of every timed mc3hip_run call
    zlag = int(ctl_opts.get('zlag', 0))

    def recycle(g):
        # every rank at the same refresh, between two barriers: nobody stores into a replica
        # while it is being rearranged.  barrier() is also the VOTE (comm.quiesce: sync on
        # every rank, all-gather of the verdicts, roll-back everywhere if a launch gave up
        # anywhere) -- mc3hip_recycle releases the checkpoint such a roll-back needs
        # (include/mc3hip.h), so it must come after it
        barrier(g)
        g.recycle(keep=1 + zlag, seed=prob['seed'] + 1 + recycles[0])
        ctl.barrier()
        recycles[0] += 1

    def advance(n, g=None, timed=False):
        g = g or gpu
        if comm is not None and g is gpu and not comm.native:
            # host-staged exchange (no peer access and no RCCL): one refresh at a time
            for _ in range(n):
                R0 = g.refresh_done
                g.run(1)
                comm.exchange(g, R0)
            return
        if args.dry_run:
            g.run(n, sync=False)
            return
        left = n
        while left:
            # (the position is kept here: asking the library would settle the stream, and a
            # launch that gave up on some rank is the next barrier's business, not this loop's)
            room = g.nzchain - done[0]
            if room < min(left, LAUNCH_REFRESHES):
                recycle(g)
                done[0] = g.refresh_done
                room = g.nzchain - done[0]
            m = min(left, room)
            g.run(m, sync=False)
            done[0] += m
            if timed:
                gpu_ms_acc[0] += g.last_run_ms()    # HIP events on the library's stream
                run_sizes.append(m)
            left -= m

    for _ in range(nwarm):
        advance(R)
    barrier()
    recycles_warm = recycles[0]
    t0 = time.perf_counter()
    for _ in range(nsteps):
        advance(R, timed=True)
    barrier()
    dt = time.perf_counter() - t0
    gpu_ms = gpu_ms_acc[0] if gpu_ms_acc[0] > 0 else gpu.last_run_ms()
    if world > 1:
        both = ctl.all_gather((dt, gpu_ms))
        dt, gpu_ms = max(b[0] for b in both), max(b[1] for b in both)

    consistent = None
    if world > 1:
        # every rank must hold the same Z rows: compare a digest of the last refreshes
        done = gpu.refresh_done
        last = min(8, done)
        Zt, _, _ = gpu.rows(gpu.M0 + (done - last)*nchains, last*nchains)
        digest = float(np.sum(Zt*np.arange(1, Zt.size + 1).reshape(Zt.shape)))
        box = ctl.all_gather(digest)
        consistent = bool(all(d == box[0] for d in box) and np.all(np.isfinite(Zt)))
    st = gpu.state()
    accept = float(st['numaccept'].sum())/(nlocal*nrefresh*args.thinning)
    chain_steps = nchains*nref_timed*args.thinning
    value = chain_steps/dt/1e6
    plan = dict(gpu.plan)
    if comm is not None:
        exchange = comm.mode           # (a recovery inside the run changes the exchange)
    recoveries = plan['recoveries'] + (comm.recoveries if comm else 0)

    # ---- N > 1: the north star's collective, and the one-GPU yardstick ---------------------
    # (run at the very end, once the line above all is ready to be printed: below)
    collective_rccl = None
    single = None
    if world > 1 and not args.dry_run:
        gpu.close()

    # roofline of the dominant (only) kernel.  Persistent path: one launch covers
    # `persistent` refreshes; per-refresh path: one launch per refresh.
    multi = world > 1 or force_comm
    cgraph = plan['comm_graph']
    per = plan['persistent'] if (not multi or exchange == 'p2p') else 0
    if per > 0:
        launches = sum(-(-m//per) for m in run_sizes) if run_sizes else -(-nref_timed//per)
        kernel = ('mc3hip::resident_async_kernel' if plan.get('owner_waves') else
                  'mc3hip::resident_tile_kernel') if plan.get('resident') else \
            {2: 'mc3hip::lean_demc_lag_kernel' if plan.get('zlag') else 'mc3hip::lean_demc_kernel',
             1: 'mc3hip::persist_fast_kernel', 0: 'mc3hip::persist_kernel',
             4: 'mc3hip::resident_tile_kernel'}[plan['fast']]
        structure = (f'persistent kernel, {per} refreshes per launch, Z rows handed '
                     'between workgroups in-launch')
    else:
        launches = nref_timed
        kernel = 'mc3hip::step_kernel'
        structure = 'one launch per refresh' + \
            ((f' + one RCCL all-gather per refresh, captured in a hipGraph of {cgraph} '
              'refreshes (step kernel, all-gather, scatter kernel each)' if cgraph else
              ' + one RCCL all-gather per refresh (eager)') if multi else
             f' (hipGraph of {plan["graph_nodes"]} nodes)')
    traffic = None
    hbm_per_step = None
    pmc = next((q for q in (os.path.join(ROOT, 'profiles', n) for n in PMC_FILES)
                if os.path.exists(q)), None)
    if pmc and args.thinning == 1 and world == 1 and args.workload == 'c2':
        with open(pmc) as f:
            rec = json.load(f)
        if rec['kernel'].startswith(kernel):
            # HBM bytes per launch from the PMC counters of the same kernel
            # (separate rocprofv3 --pmc passes, gfx950 FETCH_SIZE correction)
            hbm_per_step = rec['hbm_bytes_per_chain_step']
            traffic = hbm_per_step*nlocal*nref_timed/launches
    launch_s = gpu_ms*1e-3/launches
    bytes_per_launch = bytes_per_step*nlocal*args.thinning*nref_timed/launches
    achieved = bytes_per_launch/launch_s/1e9
    out = {
        'metric': METRIC,
        'value': value, 'unit': 'million chain-steps/sec',
        'n_gpus': world, 'steps': nsteps, 'warmup': nwarm,
        'ms_per_step': dt*1e3/nsteps, 'higher_is_better': True,
        'scaling': scaling, 'vs_baseline': None, 'dtype': 'f64',
        'data': 'dry-run (no GPU work: launcher plumbing only)' if args.dry_run else 'synthetic',
        'config': {
            'workload': wl_text,
            'nchains': nchains, 'ndata': len(prob['data']), 'thinning': args.thinning,
            'step': f'{R} Z refreshes of all {nchains} chains = {nchains*R*args.thinning} '
                    'chain-steps',
            'refreshes_per_step': R,
            'chains_per_gpu': nlocal,
            'parallelism': f'chains sharded x{world}' if world > 1 else 'single GPU',
            'collective': {None: None, 'rccl': 'one in-place RCCL all-gather of new Z rows per '
                           'refresh' + (f', replayed from a hipGraph ({cgraph} refreshes per '
                                        'replay)' if cgraph else ', eager'),
                           'p2p': 'none: every new Z row is stored peer-to-peer into '
                           'every rank\'s replica by the persistent kernel (xGMI)',
                           'host': 'host-staged'}[exchange],
            'collective_rccl': collective_rccl,
            'single_gpu_same_ensemble': single,
            'launch_structure': structure,
            'waves_per_chain': plan['waves_per_chain'],
            'acceptance_rate': accept,
            'replicas_consistent': consistent,
            'recoveries': recoveries,
            'control_plane': None if world == 1 else 'mc3_amd.parallel.SocketComm (TCP)',
            'recycles_in_timed_region': recycles[0] - recycles_warm,
            'zlag': plan.get('zlag', 0),
            'remote_delay_ns': plan.get('remote_delay_ns', 0),
            # what "the same results as the reference" means here
            'semantics': ('lockstep (Jacobi) DEMC: during refresh R a chain sees the others '
                          'through the rows of refresh R-1 (the reference updates the chains '
                          'of a worker one after another, Gauss-Seidel)' if not plan.get('zlag')
                          else
                          f'lockstep DEMC in the DECLARED MULTI-GPU CADENCE, zlag = '
                          f'{plan["zlag"]}: during refresh R a chain sees the others through '
                          f'the rows of refresh R-1-{plan["zlag"]} (the strict lockstep, '
                          'R-1, is the one-GPU default; the reference updates the chains of a '
                          'worker one after another, Gauss-Seidel) so that a row has '
                          f'{plan["zlag"]} iterations\' time to cross xGMI and an iteration '
                          'costs max(arithmetic, hop/zlag), not their sum; results depend on '
                          'zlag and not on how the chains are sharded (SURVEY.md H4)') +
                         '; draws are Philox4x32-10 keyed by (seed, iteration, chain), one '
                         'support-distribution vector per chain (the reference: MT19937, one '
                         'per worker-iteration); ncpu is ignored.  Bit-exact against '
                         'oracle/oracle_lockstep.c (which restates either cadence), whose '
                         'step function replays the reference\'s own runs from a draw tape '
                         '(tests/test_oracle_tape.py)',
        },
        'roofline': {
            # what bounds the kernel is the latency of the hand-off between workgroups and of
            # one wave's dependent FP64 chain; `frac` stays the contractual figure: the
            # algorithmic bytes per second against the HBM peak
            'bound': 'latency', 'contract_bound': 'hbm',
            'achieved': achieved, 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': achieved/HBM_PEAK_GBS, 'traffic': traffic,
            # (`traffic` and `hw` come from the committed PMC passes of the same kernel --
            # rocprofv3 --pmc cannot run inside this process -- not from this run)
            'traffic_from_profile': traffic is not None, 'measured_in_this_run': {
                'achieved': True, 'avg_launch_us': True, 'traffic': False, 'hw': False},
            'kernel': kernel, 'launches': launches,
            'algorithmic_bytes_per_launch': bytes_per_launch,
            'avg_launch_us': launch_s*1e6,
            # the honest denominators next to the contractual one: what the memory
            # system really moved, and the latency chain that really bounds an iteration
            'hw': None if hbm_per_step is None else {
                'hbm_GBs': hbm_per_step*chain_steps/world/(gpu_ms*1e-3)/1e9,
                'frac_of_peak': hbm_per_step*chain_steps/world/(gpu_ms*1e-3)/1e9/HBM_PEAK_GBS,
                'hbm_bytes_per_chain_step': hbm_per_step, 'from_profile': True,
                'source': 'profiles/' + os.path.basename(pmc)},
            'note': f'algorithmic bytes (SURVEY 8d: {bytes_per_step} B per chain-step) x '
                    'chain-steps per launch / average launch duration (HIP events on the '
                    'library stream over the timed region / launches).  The operand streams '
                    'are chain-invariant: they live in registers for a whole launch, so '
                    'hardware HBM traffic (`traffic`: bytes per launch from the PMC passes '
                    'named in hw.source, FETCH_SIZE doubled per the gfx950 note) is far below '
                    'the algorithmic figure',
        },
    }
    if args.workload == 'c5':
        flops = 2.0*10*len(prob['data'])           # per chain-step: the contraction
        useful = value*1e6*flops/1e12/world
        # the K = 10 contraction on the FP64 matrix pipe bounds this workload: ITS fraction is
        # the roofline fraction (the algorithmic bytes of the chain-invariant design matrix
        # would read ~20x the HBM peak: kept as `hbm_contract`, never as `frac`)
        out['roofline'].update({
            'hbm_contract': {'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': achieved/HBM_PEAK_GBS},
            'bound': 'mfma', 'contract_bound': 'mfma', 'achieved': useful,
            'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': useful/FP64_PEAK_TFLOPS})
        out['roofline']['fp64'] = {
            'useful_TFLOPs': useful, 'peak': FP64_PEAK_TFLOPS,
            'frac': useful/FP64_PEAK_TFLOPS,
            'note': 'design-matrix models: the K = 10 contraction on the shared FP64 matrix / '
                    'vector pipe is what bounds them, per GPU'}
    else:
        us_it = launch_s*1e6*launches/nref_timed/args.thinning
        floor_us = (0.53 + 0.42) if not plan.get('zlag') else 0.42
        # the physically meaningful fraction next to the contractual one: the latency chain
        # that bounds an iteration / what an iteration takes
        out['roofline']['latency_frac'] = floor_us/us_it
        out['roofline']['latency_model'] = {
            'us_per_iteration': us_it,
            'floor_us': floor_us,
            'floor': 'every iteration of every chain needs two rows other chains (other '
                     'CUs, other XCDs) produced in the iteration before: one cross-CU '
                     'hand-off (0.53 us per hop between two single-wave workgroups on this '
                     'chip, tools/probes/xgmi_handoff.hip run on one device, '
                     'profiles/r03_handoff_probe_one_gpu.json; MI355X_MICROARCH.md '
                     '"handoff-1to1": 0.8 us) + the dependent FP64 chain of one wave alone '
                     'on its SIMD, proposal -> chisq of 1000 points (16 per lane) -> 6 '
                     'cross-lane adds -> accept, ~1000 cycles = 0.42 us at 2.4 GHz '
                     '(tools/ticks.py).  The rest of an iteration is the second partner '
                     'row (0.27 extra polls per iteration on average: the later of two '
                     'partners decides) and the publish',
            'note': 'one wave per CU on 256 CUs = 1/4 of the SIMDs, ~2 % of the FP64 vector '
                    'peak: the workload (256 chains, 1000 points) cannot fill the chip; '
                    'it is latency-bound, neither bandwidth- nor FLOP-bound'}
    # the physically meaningful fraction under ONE name for every workload, next to the
    # contractual `frac`: what fraction of its real limiter the kernel reaches
    rl = out['roofline']
    if args.workload == 'c5':
        rl['limiter'], rl['limiter_frac'] = 'fp64 matrix / vector pipe', rl['frac']
    else:
        rl['limiter'] = 'latency of the hand-off between workgroups + one wave\'s dependent ' \
                        'FP64 chain (latency_model)'
        rl['limiter_frac'] = rl['latency_frac']
    cadence = parallel.resolve_zlag('auto', args.thinning)
    out['values'] = {
        'note': '`value` is the lockstep cadence named in config.zlag (default 0: strict, the '
                'same random process at every N -- mc3_amd.sample()\'s default on any number of '
                'ranks); `cadence` is the declared multi-GPU cadence (zlag refreshes of '
                'latency hiding, opt-in) measured in the same run; compare N GPUs with one GPU '
                'IN THE SAME CADENCE (single_gpu_same_ensemble)',
        ('strict' if not plan.get('zlag') else f'zlag{plan["zlag"]}'): {
            'zlag': plan.get('zlag', 0), 'million_chain_steps_per_s': value,
            'exchange': exchange, 'is_value': True},
        'cadence': None, 'single_gpu_same_ensemble': None}
    if world == 1:
        gpu.close()
    if rank == 0 and world == 1 and not args.dry_run and not args.no_extras:
        # one GPU in the declared cadence: the like-for-like yardstick of a sharded run's
        # `values.cadence` (and what a rank on a GPU of its own runs at)
        out['values']['cadence'] = _single_gpu(args, build_problem, nchains, local_rank, engine,
                                               cadence)
        if args.workload == 'c2':
            # the other single-GPU BASELINE configurations, driver-timed (fresh process each)
            out['configs'] = run_children(['c3one', 'c4', 'c5'])
        probe = hbm_peak_probe()
        rl['peak_measured'] = probe
        if 'GBps' in probe and rl.get('unit') == 'GB/s':
            rl['frac_of_peak_measured'] = rl['achieved']/probe['GBps']
        elif 'GBps' in probe and 'hbm_contract' in rl:
            rl['hbm_contract']['frac_of_peak_measured'] = \
                rl['hbm_contract']['achieved']/probe['GBps']
    if rank == 0 and world == 1 and not args.dry_run and not args.no_e2e:
        out['config']['e2e'] = end_to_end()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.dry_run:
        # (in a child process that never loads the HIP runtime: its worker processes are forked)
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__),
                                  '--cpu-baseline-child'], capture_output=True, text=True,
                                 timeout=300)
            out['cpu_baseline'] = json.loads(res.stdout.strip().splitlines()[-1])
        except Exception as e:      # noqa: BLE001
            out['cpu_baseline'] = {'error': (str(e) + getattr(e, 'stderr', ''))[:300]}
    import threading
    emit_lock, emitted = threading.Lock(), [False]

    def emit():
        import ctypes
        with emit_lock:                 # (the extras' timer may fire while the main thread
            if emitted[0]:              # is printing: one line, whoever comes first)
                return
            emitted[0] = True
        try:
            ctypes.CDLL(None).fflush(None)   # (what native code still holds in C stdio)
        except OSError:
            pass
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        if rank == 0:
            print(json.dumps(out), flush=True)

    if world > 1 and not args.dry_run and not args.no_extras:
        # The two extra segments have never run on more than one physical GPU: should one of
        # them not come back (RCCL's set-up across processes, say), the line with the
        # peer-to-peer figure is printed all the same and every rank leaves.
        limit = float(os.environ.get('MC3HIP_BENCH_EXTRAS_TIMEOUT', 300))

        def give_up():
            if out['config'].get('collective_rccl') is None:
                out['config']['collective_rccl'] = {
                    'error': f'no result within {limit:.0f} s; the figure above stands'}
            emit()
            # (the figure of record was printed; the hung extras are still an error worth a code)
            os._exit(0 if out['config'].get('single_gpu_same_ensemble') is not None
                     or out['config']['collective_rccl'].get('error') is None else 4)
        timer = threading.Timer(limit, give_up)
        timer.daemon = True
        timer.start()
        if not zlag:
            out['values']['cadence'] = cadence_segment(
                args, ctl, engine, build_problem, nchains, nlocal, chain0, local_rank, cadence,
                shared_device)
            ctl.barrier()
        out['config']['collective_rccl'] = rccl_segment(
            args, ctl, make_engine, build_problem, nchains, nlocal, chain0, local_rank, engine)
        ctl.barrier()
        if rank == 0:
            # ONE GPU, the whole ensemble, in the strict lockstep and in the cadence: N GPUs
            # against one in the SAME random process, either way
            single = single_gpu_segment(args, build_problem, nchains, local_rank, engine,
                                        zlag or cadence)
            out['config']['single_gpu_same_ensemble'] = single
            out['values']['single_gpu_same_ensemble'] = single
        ctl.barrier()
        timer.cancel()
    ctl.barrier()
    ctl.close()
    emit()


def cadence_segment(args, ctl, engine, build_problem, nchains, nlocal, chain0, device, zlag,
                    shared_device):
    """The same sharded ensemble in the DECLARED CADENCE (the other chains seen `zlag` refreshes
    late, so that a row has that long to cross xGMI): a second engine per rank, the same
    exchange set-up by consensus, timed on its own.  `values.cadence`."""
    if args.workload == 'c5':
        nwarm, ntimed = 16, 100
    else:
        nwarm, ntimed = (4096, 16384) if shared_device else (16384, 4*LAUNCH_REFRESHES)
    prob = build_problem(nchains, nwarm + ntimed + 8, args.thinning)
    saved = (ctl.native, ctl.mode, ctl.wants_p2p)
    g, ok, err = None, True, None
    try:
        ctl.wants_p2p = not os.environ.get('MC3HIP_BENCH_NO_P2P')
        opts = ctl.engine_options(device, args.thinning, zlag)
        g = engine.DeviceChains(device=device, chain0=chain0, nlocal=nlocal, **opts, **prob)
        Z0, lp0 = initial_population(prob, lambda Z: g.eval_chisq(g.expand_free(Z)))
        g.set_population(Z0, lp0)
    except Exception as e:      # noqa: BLE001
        ok, err = False, str(e)[:300]
    out = None
    if ctl._all_ok(ok):
        ctl.setup(g)
        if not ctl.native:
            out = {'error': 'neither the peer-to-peer hand-off nor RCCL on every rank'}
        else:
            mode = ctl.mode
            g.run(nwarm, sync=False)
            ctl.quiesce(g)
            g.sync()
            ctl.barrier()
            t0 = time.perf_counter()
            g.run(ntimed, sync=False)
            ctl.quiesce(g)
            g.sync()
            ctl.barrier()
            dt = max(ctl.all_gather(time.perf_counter() - t0))
            out = {'zlag': int(g.plan.get('zlag', 0)), 'exchange': ctl.mode,
                   'exchange_at_start': mode, 'refreshes_timed': ntimed,
                   'us_per_refresh': dt/ntimed*1e6,
                   'million_chain_steps_per_s': nchains*ntimed*args.thinning/dt/1e6,
                   'recoveries': g.plan['recoveries'] + ctl.recoveries, 'is_value': False}
    else:
        out = {'error': err or 'an engine could not be created on every rank'}
    if g is not None:
        g.close()
    ctl.native, ctl.mode, ctl.wants_p2p = saved
    return out


def rccl_segment(args, ctl, make_engine, build_problem, nchains, nlocal, chain0, device, engine):
    """The same ensemble through ONE in-place RCCL all-gather of the new Z rows per refresh
    (libmc3hip calls RCCL itself; replayed from a hipGraph), timed on its own: the north
    star's collective, with RCCL's own word on how many ranks it spanned."""
    nwarm, ntimed = (8, 24) if args.workload == 'c5' else (2048, 8192)
    prob = build_problem(nchains, nwarm + ntimed + 8, args.thinning)
    g = None
    ok, err = True, None
    try:
        g = engine.DeviceChains(device=device, chain0=chain0, nlocal=nlocal, **prob)
        Z0, lp0 = initial_population(prob, lambda Z: g.eval_chisq(g.expand_free(Z)))
        g.set_population(Z0, lp0)
    except Exception as e:      # noqa: BLE001
        ok, err = False, str(e)
    saved = (ctl.native, ctl.mode)
    if ctl._all_ok(ok):
        ctl._setup_rccl(g)
        ok = ctl.mode == 'rccl'
        if not ok:
            err = 'RCCL communicator could not be set up on every rank (RCCL refuses ' \
                  'several ranks on one device)'
    else:
        ok = False
    ctl.native, ctl.mode = saved
    out = None
    if ok:
        g.run(nwarm)
        ctl.barrier()
        t0 = time.perf_counter()
        g.run(ntimed)
        ctl.barrier()
        dt = max(ctl.all_gather(time.perf_counter() - t0))
        nr, _ = g.comm_info()
        plan = g.plan
        out = {'exchange': 'one in-place ncclAllGather of the refresh\'s Z rows per refresh',
               'rccl_ranks': nr, 'refreshes_timed': ntimed,
               'us_per_refresh': dt/ntimed*1e6,
               'million_chain_steps_per_s': nchains*ntimed*args.thinning/dt/1e6,
               'hipgraph_refreshes_per_replay': plan['comm_graph'],
               'payload_bytes_per_rank': nlocal*128}
    else:
        out = {'error': err or 'not available on every rank'}
    if g is not None:
        g.close()
    return out


def single_gpu_segment(args, build_problem, nchains, device, engine, zlag=0):
    """The whole ensemble in ONE context on one GPU (no exchange at all): in the strict
    lockstep, and -- `same_cadence` -- in the cadence the sharded run used (the like-for-like
    yardstick: same semantics, one GPU against N)."""
    out = _single_gpu(args, build_problem, nchains, device, engine, 0)
    if zlag:
        out['same_cadence'] = _single_gpu(args, build_problem, nchains, device, engine, zlag)
    return out


def _single_gpu(args, build_problem, nchains, device, engine, zlag):
    # (long enough for the clocks to come back up after the idle seconds before it: with 8192
    # refreshes -- 10 ms -- the figure of one and the same kernel varied by a factor of two)
    nwarm, ntimed = (40, 200) if args.workload == 'c5' else (16384, 65536)
    prob = build_problem(nchains, nwarm + ntimed + 8, args.thinning)
    try:
        with engine.DeviceChains(device=device, zlag=zlag, **prob) as g:
            Z0, lp0 = initial_population(prob, lambda Z: g.eval_chisq(g.expand_free(Z)))
            g.set_population(Z0, lp0)
            g.run(nwarm)
            t0 = time.perf_counter()
            g.run(ntimed)
            dt = time.perf_counter() - t0
            plan = g.plan
        return {'nchains': nchains, 'zlag': zlag, 'million_chain_steps_per_s':
                nchains*ntimed*args.thinning/dt/1e6, 'us_per_refresh': dt/ntimed*1e6,
                'persistent': plan['persistent'], 'refreshes_timed': ntimed,
                'chains_per_workgroup': plan.get('chains_per_workgroup', 1)}
    except Exception as e:      # noqa: BLE001
        return {'error': str(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=12)         # timed batches of Z refreshes
    ap.add_argument('--warmup', type=int, default=2)         # untimed batches of the same size
    ap.add_argument('--refreshes', type=int, default=0,
                    help='Z refreshes per step (default: up to 6 persistent launches of 16384, '
                         'as many as (steps+warmup) batches of Z fit the memory budget)')
    ap.add_argument('--thinning', type=int, default=1)
    ap.add_argument('--zlag', default=None,
                    help='the lockstep cadence of `value`: the other chains are seen that many '
                         'refreshes late (an integer or "auto" = parallel.CADENCE_ZLAG; default '
                         '0 = strict lockstep at every N; the cadence is measured beside it, '
                         '`values.cadence`)')
    ap.add_argument('--config-child', default=None, choices=sorted(CONFIGS),
                    help='(internal) one of the other BASELINE configurations, one JSON line')
    ap.add_argument('--cpu-baseline-child', action='store_true',
                    help='(internal) the CPU baseline, one JSON line; never touches the GPU')
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='c2',
                    help='c2: the headline, 256 chains per GPU (weak scaling); c3 / c5: BASELINE '
                         'configs 3 and 5, a fixed ensemble split over the GPUs (strong scaling)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the extra segments (several ranks: the cadence, the RCCL route, '
                         'the one-GPU yardstick; one GPU: the cadence, the other BASELINE '
                         'configurations, the HBM probe) -- sweeps and profiler passes')
    ap.add_argument('--e2e-child', default=None, choices=E2E_CASES,
                    help='(internal) one end-to-end sample() case in this process')
    ap.add_argument('--no-e2e', action='store_true',
                    help='skip the end-to-end sample() timings (config.e2e)')
    ap.add_argument('--dry-run', action='store_true',
                    help='no GPU work: exercises the launcher / rendezvous / reporting only')
    args = ap.parse_args()
    if args.cpu_baseline_child:
        print(json.dumps(cpu_baseline()), flush=True)
        return
    if args.e2e_child:
        end_to_end_child(args.e2e_child)
        return
    if args.config_child:
        config_child(args.config_child)
        return
    if args.zlag is not None and args.zlag != 'auto':
        args.zlag = int(args.zlag)
    if 'RANK' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and not (world == 1 and args.gpus <= 1):
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}',
              file=sys.stderr)
    run_rank(args)


if __name__ == '__main__':
    main()
