#!/usr/bin/env python3
"""bench.py -- million chain-steps/sec of the MI355X chain-update path.

Workload (BASELINE.json configs[1], "C2"): quadratic 3-parameter fit,
ndata=1000, 256 chains per GPU, sampler='demc', thinning=1, fepsilon=0,
synthetic data as in examples/tutorial.py:22-46 of the reference (SURVEY.md
section 8d).  A "step" is one batch of `--refreshes` Z refreshes (default 16 384 =
one launch of the persistent kernel, 4.19 M chain-steps per GPU): in every refresh
every chain of the ensemble advances `thinning` iterations (proposal -> model ->
chisq -> Metropolis -> Z append).  Whatever --steps/--warmup are given, a step is
long against launch and synchronisation overheads; the batch shrinks only when Z
(128 B per chain and refresh) would not fit a 24 GiB budget.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1 is weak scaling: 256*N chains, rank r owns chains [256 r, 256 (r+1)); the
only data-path collective is one in-place RCCL all-gather of the refresh's new
Z rows (libmc3hip calls RCCL itself; torch.distributed/gloo is only used to
hand out the RCCL id, for the barriers and for the max-over-ranks time).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHAINS_PER_GPU = 256
# the metric exactly as BASELINE.json names it
METRIC = 'million chain-steps/sec (nchains\u00d7iters/s), 256-chain quad fit ndata=1000'
NDATA = 1000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes per chain-step, Gaussian poly, nfree=3:
#   8*S*ndata + 8*nfree*(R+1) + 8 with S=3 streams (x, data, uncert), R=3 rows
BYTES_PER_CHAIN_STEP = 8*3*NDATA + 8*3*(3 + 1) + 8


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


def initial_population(prob, eval_chisq):
    """M0 = hsize*nchains in-bounds draws around the initial guess
    (mcmc_driver.py:229-262), chisq from the device."""
    rs = np.random.RandomState(5)
    M0 = prob['hsize']*prob['nchains']
    x0, sig = prob['params'], prob['pstep']
    Z0 = np.empty((0, 3))
    while len(Z0) < M0:
        trial = rs.normal(x0, sig, (2*M0, 3))
        ok = np.all((trial >= prob['pmin']) & (trial <= prob['pmax']), axis=1)
        Z0 = np.concatenate([Z0, trial[ok]])
    Z0 = np.ascontiguousarray(Z0[:M0])
    return Z0, -0.5*eval_chisq(Z0)


def cpu_baseline(target_seconds=12.0):
    """The oracle (oracle/oracle_lockstep.c: the same lockstep chain update in
    plain C, OpenMP over chains) on this box's host cores, same C2 workload,
    bounded to ~target_seconds.  The thread count is the fastest of a short
    calibration over 1..usable cores (a container may see more CPUs than it is
    allowed to use)."""
    from oracle import lockstep as ls
    usable = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') \
        else (os.cpu_count() or 1)
    prob = c2_problem(CHAINS_PER_GPU, 400000)
    orc = ls.LockstepRun(**prob)
    Z0, lp0 = initial_population(prob, lambda Z: orc.eval_chisq(Z))
    orc.set_population(Z0, lp0)

    def rate(nthreads, n):
        t0 = time.perf_counter()
        orc.run(nrefresh=n, nthreads=nthreads)
        return n/(time.perf_counter() - t0)

    rate(1, 20)
    best_threads, best_rate = 1, rate(1, 100)
    for nt in (2, 4, 8, 16, 32, 64, 128, 256):
        if nt > min(usable, CHAINS_PER_GPU):
            break
        r = rate(nt, 200)
        if r > best_rate:
            best_threads, best_rate = nt, r
        elif r < 0.5*best_rate:
            break
    chunk = int(max(100, best_rate))            # about one second of work
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < target_seconds and n + chunk <= 390000:
        orc.run(nrefresh=chunk, nthreads=best_threads)
        n += chunk
    dt = time.perf_counter() - t0
    return {
        'value': CHAINS_PER_GPU*n/dt/1e6, 'unit': 'million chain-steps/sec',
        'cores': best_threads, 'kind': 'port',
        'sample': f'{n} iterations of the same 256-chain C2 workload '
                  f'({dt:.1f} s), oracle_lockstep.c, {best_threads} OpenMP threads '
                  f'(best of a calibration up to {usable} usable CPUs)',
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=12)         # timed batches of Z refreshes
    ap.add_argument('--warmup', type=int, default=2)         # untimed batches of the same size
    ap.add_argument('--refreshes', type=int, default=0,
                    help='Z refreshes per step (default 16384 = one persistent launch; '
                         'fewer when (steps+warmup) batches of Z would exceed 24 GiB)')
    ap.add_argument('--thinning', type=int, default=1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    # stdout carries ONE JSON line: whatever native libraries print there meanwhile (gloo's
    # and RCCL's banners) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'MC3HIP_BENCH_DEVICE' in os.environ:      # testing: several ranks on one GPU
        local_rank = int(os.environ['MC3HIP_BENCH_DEVICE'])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus N>1 must be launched with torch.distributed.run '
                     '(one process per GPU)')
        args.gpus = world

    # MC3HIP_BENCH_FORCE_COMM=1 drives the multi-rank code path (gloo group, RCCL
    # communicator, one all-gather per refresh) with a single rank, for testing
    force_comm = bool(os.environ.get('MC3HIP_BENCH_FORCE_COMM')) and world == 1
    dist = None
    if world > 1 or force_comm:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        dist.init_process_group('gloo', rank=rank, world_size=world)

    from mc3_amd import engine, parallel

    if world > 1:
        # a multi-rank run that stops making progress must not hang the node: dump every
        # thread's stack and exit after MC3HIP_BENCH_WATCHDOG seconds (default 900)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get('MC3HIP_BENCH_WATCHDOG', 900)),
                                          exit=True)

    nchains = CHAINS_PER_GPU*world
    nbatch = max(1, args.steps) + max(0, args.warmup)
    row_bytes = nchains*(16*8 + 8 + 4)            # padded Z row + log_post + zchain
    R = args.refreshes or int(max(64, min(16384, (24 << 30)//(row_bytes*nbatch))))
    nref_timed, nref_warm = max(1, args.steps)*R, max(0, args.warmup)*R
    nrefresh = nref_timed + nref_warm
    prob = c2_problem(nchains, nrefresh + 64, args.thinning)
    comm = None
    exchange = None

    def make_engine(p2p):
        g = engine.DeviceChains(device=local_rank, chain0=rank*CHAINS_PER_GPU,
                                nlocal=CHAINS_PER_GPU, p2p=p2p, **prob)
        Z0, lp0 = initial_population(prob, g.eval_chisq)
        g.set_population(Z0, lp0)
        return g

    if world > 1:
        # preferred: the in-launch hand-off across GPUs (peer-to-peer stores into every
        # rank's Z replica); validated on a short run, else one RCCL all-gather per
        # refresh.  Every rank takes the same branch.
        comm = parallel.TorchDistComm(use_p2p=not os.environ.get('MC3HIP_BENCH_NO_P2P'))
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
                comm = parallel.TorchDistComm(use_p2p=False)
                gpu = make_engine(False)
                comm.setup(gpu)
        exchange = comm.mode
    else:
        gpu = make_engine(False)
        if force_comm:
            ids = [engine.comm_unique_id()]
            gpu.comm_init(ids[0], 0, 1)
            exchange = 'rccl'

    def barrier():
        # multi-rank: quiesce() drains the library's stream, lets the ranks agree that
        # the segment completed everywhere (peer-to-peer stores: every peer has finished
        # too) and, if a launch gave up anywhere, rolls all of them back and redoes it
        if comm is not None:
            comm.quiesce(gpu)
        gpu.sync()
        if world > 1 or force_comm:
            dist.barrier()

    def advance(n):
        if comm is not None and not comm.native:
            # host-staged exchange (no peer access and no RCCL): one refresh at a time
            for _ in range(n):
                R = gpu.refresh_done
                gpu.run(1)
                comm.exchange(gpu, R)
        else:
            gpu.run(n, sync=False)

    if nref_warm:
        advance(nref_warm)
    barrier()
    t0 = time.perf_counter()
    advance(nref_timed)
    barrier()
    dt = time.perf_counter() - t0
    gpu_ms = gpu.last_run_ms()          # HIP events on the library's stream
    if world > 1 or force_comm:
        t = torch.tensor([dt, gpu_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, gpu_ms = float(t[0]), float(t[1])

    consistent = None
    if world > 1:
        # every rank must hold the same Z rows: compare a digest of the last refreshes
        done = gpu.refresh_done
        Zt, _, _ = gpu.rows(gpu.M0 + (done - 8)*nchains, 8*nchains)
        digest = float(np.sum(Zt*np.arange(1, Zt.size + 1).reshape(Zt.shape)))
        box = [None]*world
        dist.all_gather_object(box, digest)
        consistent = bool(all(d == box[0] for d in box) and np.all(np.isfinite(Zt)))
    st = gpu.state()
    accept = float(st['numaccept'].sum())/(CHAINS_PER_GPU*gpu.refresh_done*args.thinning)
    chain_steps = nchains*nref_timed*args.thinning
    value = chain_steps/dt/1e6

    # roofline of the dominant (only) kernel.  Persistent path: one launch covers
    # `persistent` refreshes; per-refresh path: one launch per refresh.
    multi = world > 1 or force_comm
    if comm is not None:
        exchange = comm.mode           # (a recovery inside the run changes the exchange)
    cgraph = gpu.plan['comm_graph']
    per = gpu.plan['persistent'] if (not multi or exchange == 'p2p') else 0
    if per > 0:
        launches = -(-nref_timed//per)
        kernel = {2: 'mc3hip::lean_demc_kernel', 1: 'mc3hip::persist_fast_kernel',
                  0: 'mc3hip::persist_kernel'}[gpu.plan['fast']]
        structure = (f'persistent kernel, {per} refreshes per launch, Z rows handed '
                     'between workgroups in-launch')
    else:
        launches = nref_timed
        kernel = 'mc3hip::step_kernel'
        structure = 'one launch per refresh' + \
            ((f' + one RCCL all-gather per refresh, captured in a hipGraph of {cgraph} '
              'refreshes (step kernel, all-gather, scatter kernel each)' if cgraph else
              ' + one RCCL all-gather per refresh (eager)') if multi else
             f' (hipGraph of {gpu.plan["graph_nodes"]} nodes)')
    traffic = None
    hbm_per_step = None
    pmc = next((q for q in (os.path.join(ROOT, 'profiles', n) for n in
                            ('r02_c2_pmc.json', 'r01g_c2_pmc.json')) if os.path.exists(q)), None)
    if pmc and args.thinning == 1 and world == 1:
        with open(pmc) as f:
            rec = json.load(f)
        if rec['kernel'].startswith(kernel):
            # HBM bytes per launch from the PMC counters of the same kernel
            # (separate rocprofv3 --pmc passes, gfx950 FETCH_SIZE correction)
            hbm_per_step = rec['hbm_bytes_per_chain_step']
            traffic = hbm_per_step*CHAINS_PER_GPU*nref_timed/launches
    launch_s = gpu_ms*1e-3/launches
    bytes_per_launch = BYTES_PER_CHAIN_STEP*CHAINS_PER_GPU*args.thinning*nref_timed/launches
    achieved = bytes_per_launch/launch_s/1e9
    out = {
        'metric': METRIC,
        'value': value, 'unit': 'million chain-steps/sec',
        'n_gpus': world, 'steps': max(1, args.steps), 'warmup': max(0, args.warmup),
        'ms_per_step': dt*1e3/max(1, args.steps), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
        'data': 'synthetic',
        'config': {
            'workload': 'C2: quadratic 3-param fit, ndata=1000, 256 chains/GPU, '
                        "sampler='demc', fepsilon=0",
            'nchains': nchains, 'ndata': NDATA, 'thinning': args.thinning,
            'step': f'{R} Z refreshes of all {nchains} chains = {nchains*R*args.thinning} '
                    'chain-steps',
            'refreshes_per_step': R,
            'chains_per_gpu': CHAINS_PER_GPU,
            'parallelism': f'chains sharded x{world}' if world > 1 else 'single GPU',
            'collective': {None: None, 'rccl': 'one in-place RCCL all-gather of new Z rows per '
                           'refresh' + (f', replayed from a hipGraph ({cgraph} refreshes per '
                                        'replay)' if cgraph else ', eager'), 'p2p': 'none: every new Z row is stored peer-to-peer into '
                           'every rank\'s replica by the persistent kernel (xGMI)',
                           'host': 'host-staged'}[exchange],
            'launch_structure': structure,
            'waves_per_chain': gpu.plan['waves_per_chain'],
            'acceptance_rate': accept,
            'replicas_consistent': consistent,
            'recoveries': gpu.plan['recoveries'] + (comm.recoveries if comm else 0),
            # what "the same results as the reference" means here
            'semantics': 'lockstep (Jacobi) DEMC: during refresh R a chain sees the others '
                         'through the rows of refresh R-1 (the reference updates the chains '
                         'of a worker one after another, Gauss-Seidel); draws are '
                         'Philox4x32-10 keyed by (seed, iteration, chain), one support-'
                         'distribution vector per chain (the reference: MT19937, one per '
                         'worker-iteration); ncpu is ignored.  Bit-exact against '
                         'oracle/oracle_lockstep.c, whose step function replays the '
                         'reference\'s own runs from a draw tape (tests/test_oracle_tape.py)',
        },
        'roofline': {
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': achieved/HBM_PEAK_GBS, 'traffic': traffic,
            'kernel': kernel, 'launches': launches,
            'algorithmic_bytes_per_launch': bytes_per_launch,
            'avg_launch_us': launch_s*1e6,
            # the honest denominators next to the contractual one: what the memory
            # system really moved, and the latency chain that really bounds an iteration
            'hw': None if hbm_per_step is None else {
                'hbm_GBs': hbm_per_step*chain_steps/world/(gpu_ms*1e-3)/1e9,
                'frac_of_peak': hbm_per_step*chain_steps/world/(gpu_ms*1e-3)/1e9/HBM_PEAK_GBS,
                'hbm_bytes_per_chain_step': hbm_per_step,
                'source': os.path.basename(pmc)},
            'latency_model': {
                'us_per_iteration': launch_s*1e6*launches/nref_timed/args.thinning,
                'floor_us': 0.8 + 0.42,
                'floor': 'every iteration of every chain needs two rows other chains (other '
                         'CUs, other XCDs) produced in the iteration before: one cross-CU '
                         'hand-off (MI355X_MICROARCH.md "handoff-1to1" 0.8 us) + the dependent '
                         'FP64 chain of one wave alone on its SIMD, proposal -> chisq of 1000 '
                         'points (16 per lane) -> 6 cross-lane adds -> exp -> accept, '
                         '~1000 cycles = 0.42 us at 2.4 GHz (tools/ticks.py)',
                'note': 'one wave per CU on 256 CUs = 1/4 of the SIMDs, ~2 % of the FP64 vector '
                        'peak: the workload (256 chains, 1000 points) cannot fill the chip; '
                        'it is latency-bound, neither bandwidth- nor FLOP-bound'},
            'note': 'algorithmic bytes (SURVEY 8d: 24104 B per chain-step) x chain-steps '
                    'per launch / average launch duration (HIP events on the library '
                    'stream over the timed region / launches). x/data/uncert are '
                    'chain-invariant: they live in registers for a whole launch, so '
                    'hardware HBM traffic (traffic: bytes per launch from '
                    'profiles/r01g_c2_pmc.json, FETCH_SIZE doubled per the gfx950 note) '
                    'is far below the algorithmic figure',
        },
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
    gpu.close()
    if world > 1 or force_comm:
        dist.barrier()
        dist.destroy_process_group()
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)       # (what native code still holds in C stdio)
    except OSError:
        pass
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
