"""oracle/lockstep.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front end of oracle_lockstep.c: the reference's chain-update rules
(mc3/chain.py:183-299) in the lockstep / Philox form the HIP kernels implement.
This is the bit-exact checker for the GPU path: same seed, same inputs =>
identical Z rows, chain ids, accept and out-of-bounds counters, chisq values.
"""
import ctypes

import numpy as np

from . import kernels as ok

SAMPLERS = {'mrw': 0, 'demc': 1, 'snooker': 2}
MODELS = {'poly': 0, 'linear': 1}

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_lp = ctypes.POINTER(ctypes.c_int64)


class _Problem(ctypes.Structure):
    _fields_ = [
        ('nchains', ctypes.c_int32), ('chain0', ctypes.c_int32),
        ('nlocal', ctypes.c_int32),
        ('nparams', ctypes.c_int32), ('nfree', ctypes.c_int32),
        ('nshare', ctypes.c_int32), ('ndata', ctypes.c_int32),
        ('sampler', ctypes.c_int32), ('wlike', ctypes.c_int32),
        ('model_kind', ctypes.c_int32), ('nmodel', ctypes.c_int32),
        ('thinning', ctypes.c_int32),
        ('ngprior', ctypes.c_int32),
        ('tile', ctypes.c_int32), ('vlanes', ctypes.c_int32),
        ('wl_threads', ctypes.c_int32),
        ('tile_tree', ctypes.c_int32), ('zlag', ctypes.c_int32),
        ('wl_nreal', ctypes.c_int32), ('reserved1', ctypes.c_int32),
        ('M0', ctypes.c_int64), ('zlen', ctypes.c_int64),
        ('seed', ctypes.c_uint64),
        ('gamma', ctypes.c_double), ('fepsilon', ctypes.c_double),
        ('ifree', _ip), ('ishare', _ip), ('share_src', _ip), ('igprior', _ip),
        ('params', _dp), ('pmin', _dp), ('pmax', _dp), ('pstep', _dp),
        ('prior', _dp), ('priorlow', _dp), ('priorup', _dp),
        ('indep', _dp), ('data', _dp), ('w', _dp),
    ]


class _State(ctypes.Structure):
    _fields_ = [
        ('Z', _dp), ('log_post', _dp), ('zchain', _ip),
        ('X', _dp), ('chisq', _dp),
        ('numaccept', _lp), ('outbounds', _lp),
        ('best_chisq', _dp), ('best_iter', _lp), ('bestx', _dp),
        ('trace_chisq', _dp), ('trace_flag', _ip),
        ('snap0', _dp),
        ('refresh_done', ctypes.c_int64),
    ]


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class LockstepRun:
    """One lockstep run on the CPU.

    indep: x[ndata] for model='poly' (coefficients = params[0:nmodel]),
           A[nmodel, ndata] for model='linear' (model = sum_k A[k]*params[k]).
    tile/vlanes/wl_threads: the data-sum order ("reduction plan") of the kernel this
    run is checking; the HIP library reports the plan it used (wl_threads = 64 x
    waves per chain on the wavelet path).
    """

    def __init__(self, data, uncert, indep, params, pstep, pmin=None, pmax=None,
                 prior=None, priorlow=None, priorup=None, nchains=7,
                 nsamples=None, sampler='snooker', wlike=False, thinning=1,
                 fgamma=1.0, fepsilon=0.0, hsize=10, model='poly', nmodel=None,
                 seed=0, tile=0, vlanes=None, chain0=0, nlocal=None, M0=None,
                 wl_threads=0, tile_tree=0, zlag=0, wl_nreal=0, shared=False):
        # shared: the state arrays live in anonymous MAP_SHARED memory, so that worker
        # processes forked later (run_processes) advance ONE run
        self._maps = []
        L = ok.lib()
        L.ls_run.restype = ctypes.c_int
        L.ls_eval_chisq.restype = ctypes.c_int
        self.L = L
        self.params = _f(params)
        npar = len(self.params)
        self.pstep = _f(pstep)
        self.pmin = _f(np.tile(-np.inf, npar) if pmin is None else pmin)
        self.pmax = _f(np.tile(np.inf, npar) if pmax is None else pmax)
        zeros = np.zeros(npar)
        self.prior = _f(zeros if prior is None else prior)
        self.priorlow = _f(zeros if priorlow is None else priorlow)
        self.priorup = _f(zeros if priorup is None else priorup)
        self.data = _f(data)
        self.w = _f(1.0/np.asarray(uncert, float))
        self.indep = _f(indep)
        self.ndata = len(self.data)
        self.ifree = _i(np.where(self.pstep > 0)[0])
        self.ishare = _i(np.where(self.pstep < 0)[0])
        # chain.py:247: nextp[s] = nextp[-int(pstep[s])-1]
        self.share_src = _i([(-int(self.pstep[s]) - 1) % npar for s in self.ishare])
        self.igprior = _i(np.where((self.priorlow > 0) & (self.priorup > 0))[0])
        self.nfree = len(self.ifree)
        self.nchains = int(nchains)
        self.thinning = int(thinning)
        self.sampler = sampler
        self.wlike = bool(wlike)
        if nmodel is None:
            nmodel = npar - 3 if wlike else npar
        self.nmodel = int(nmodel)
        if vlanes is None:
            vlanes = 1024
        self.hsize = int(hsize)
        self.M0 = self.hsize*self.nchains if M0 is None else int(M0)
        self.nzchain = int(np.ceil(nsamples/self.nchains/self.thinning))
        self.zlen = self.M0 + self.nzchain*self.nchains
        nf = self.nfree

        def alloc(shape, dtype=np.float64, fill=0):
            if not shared:
                return np.full(shape, fill, dtype)
            import mmap
            nbytes = max(1, int(np.prod(shape))*np.dtype(dtype).itemsize)
            m = mmap.mmap(-1, nbytes)         # (anonymous, MAP_SHARED: inherited by fork)
            self._maps.append(m)
            a = np.frombuffer(m, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
            if fill != 0:
                a[...] = fill
            return a
        self.Z = alloc((self.zlen, nf))
        self.log_post = alloc(self.zlen)
        self.zchain = alloc(self.zlen, np.int32, -1)
        self.X = alloc((self.nchains, nf))
        self.chisq = alloc(self.nchains)
        self.numaccept = alloc(self.nchains, np.int64)
        self.outbounds = alloc(nf, np.int64)
        self.best_chisq = alloc(self.nchains, fill=np.inf)
        self.best_iter = alloc(self.nchains, np.int64, -1)
        self.bestx = alloc((self.nchains, nf))
        self.snap0 = alloc((self.nchains, nf))
        self._barrier = alloc(8, np.int32) if shared else None
        self.trace_chisq = None
        self.trace_flag = None
        # gamma as the host computes it (chain.py:175)
        self.gamma = float(fgamma*2.38/np.sqrt(2*nf))

        P = _Problem()
        P.nchains = self.nchains
        P.chain0 = int(chain0)
        P.nlocal = self.nchains - int(chain0) if nlocal is None else int(nlocal)
        P.nparams = npar
        P.nfree = nf
        P.nshare = len(self.ishare)
        P.ndata = self.ndata
        P.sampler = SAMPLERS[sampler]
        P.wlike = int(self.wlike)
        P.model_kind = MODELS[model]
        P.nmodel = self.nmodel
        P.thinning = self.thinning
        P.ngprior = len(self.igprior)
        P.tile = int(tile)
        P.vlanes = int(vlanes)
        P.wl_threads = int(wl_threads)      # 0: the library's default geometry
        P.tile_tree = int(tile_tree)
        P.zlag = int(zlag)           # declared multi-GPU cadence (oracle_lockstep.c ls_run)
        P.wl_nreal = int(wl_nreal)   # wavelet series padded to 2^M: the points that are data
        self.zlag = int(zlag)
        P.M0 = self.M0
        P.zlen = self.zlen
        P.seed = int(seed)
        P.gamma = self.gamma
        P.fepsilon = float(fepsilon)
        for name in ('ifree', 'ishare', 'share_src', 'igprior'):
            setattr(P, name, getattr(self, name).ctypes.data_as(_ip))
        for name in ('params', 'pmin', 'pmax', 'pstep', 'prior', 'priorlow',
                     'priorup', 'indep', 'data', 'w'):
            setattr(P, name, getattr(self, name).ctypes.data_as(_dp))
        self.P = P
        S = _State()
        S.refresh_done = 0
        self.S = S
        self._bind()

    def _bind(self):
        S = self.S
        S.Z = self.Z.ctypes.data_as(_dp)
        S.log_post = self.log_post.ctypes.data_as(_dp)
        S.zchain = self.zchain.ctypes.data_as(_ip)
        S.X = self.X.ctypes.data_as(_dp)
        S.chisq = self.chisq.ctypes.data_as(_dp)
        S.numaccept = self.numaccept.ctypes.data_as(_lp)
        S.outbounds = self.outbounds.ctypes.data_as(_lp)
        S.best_chisq = self.best_chisq.ctypes.data_as(_dp)
        S.best_iter = self.best_iter.ctypes.data_as(_lp)
        S.bestx = self.bestx.ctypes.data_as(_dp)
        S.snap0 = self.snap0.ctypes.data_as(_dp)

    def set_population(self, Z0, log_post0):
        """Initial Z rows (mcmc_driver.py:260-261); chains start from the first
        nchains of them (chain.py:163-170)."""
        Z0 = _f(Z0)
        self.Z[0:self.M0] = Z0
        self.log_post[0:self.M0] = log_post0
        self.X[:] = Z0[0:self.nchains]
        self.snap0[:] = Z0[0:self.nchains]
        self.chisq[:] = -2.0*np.asarray(log_post0)[0:self.nchains]

    def init_population(self, kickoff='normal', seed=None):
        """mcmc_driver.py:229-270 with Philox trial draws (ls_init_population):
        (Z0, log_post0, filled, failed)."""
        Z0 = np.zeros((self.M0, self.nfree))
        lp0 = np.zeros(self.M0)
        filled, failed = ctypes.c_int64(), ctypes.c_int64()
        self.L.ls_init_population.restype = ctypes.c_int
        self.L.ls_init_population(
            ctypes.byref(self.P), ctypes.c_int(int(kickoff == 'uniform')),
            ctypes.c_uint64(int(self.P.seed if seed is None else seed)),
            Z0.ctypes.data_as(_dp), lp0.ctypes.data_as(_dp), ctypes.byref(filled),
            ctypes.byref(failed))
        return Z0, lp0, int(filled.value), int(failed.value)

    def init_population_tape(self, tape):
        """The same acceptance rule (one C function, init_population_core) fed a tape of trial
        vectors [ntrial, nfree] -- the reference's own np.random draws: (Z0, log_post0,
        filled, failed, trials consumed)."""
        tape = _f(np.atleast_2d(tape))
        Z0 = np.zeros((self.M0, self.nfree))
        lp0 = np.zeros(self.M0)
        filled, failed, used = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self.L.ls_init_population_tape.restype = ctypes.c_int
        self.L.ls_init_population_tape(
            ctypes.byref(self.P), tape.ctypes.data_as(_dp), ctypes.c_int64(len(tape)),
            Z0.ctypes.data_as(_dp), lp0.ctypes.data_as(_dp), ctypes.byref(filled),
            ctypes.byref(failed), ctypes.byref(used))
        return Z0, lp0, int(filled.value), int(failed.value), int(used.value)

    def eval_chisq(self, params):
        params = _f(np.atleast_2d(params))
        out = np.empty(len(params))
        self.L.ls_eval_chisq(ctypes.byref(self.P), params.ctypes.data_as(_dp),
                             ctypes.c_int(len(params)), out.ctypes.data_as(_dp))
        return out

    def eval_model(self, params):
        out = np.empty(self.ndata)
        self.L.ls_eval_model(ctypes.byref(self.P), _f(params).ctypes.data_as(_dp),
                             out.ctypes.data_as(_dp))
        return out

    def run(self, nrefresh=None, trace=False, nthreads=1):
        if nrefresh is None:
            nrefresh = self.nzchain - self.S.refresh_done
        if trace:
            self.trace_chisq = np.full((nrefresh*self.thinning, self.nchains), np.nan)
            self.trace_flag = np.zeros((nrefresh*self.thinning, self.nchains), np.int32)
            self.S.trace_chisq = self.trace_chisq.ctypes.data_as(_dp)
            self.S.trace_flag = self.trace_flag.ctypes.data_as(_ip)
        else:
            self.S.trace_chisq = None
            self.S.trace_flag = None
        rc = self.L.ls_run(ctypes.byref(self.P), ctypes.byref(self.S),
                           ctypes.c_int64(nrefresh), ctypes.c_int(nthreads))
        if rc != 0:
            raise RuntimeError(f'ls_run failed ({rc})')
        return self


    def run_processes(self, nrefresh, ncpu):
        """The same refreshes by `ncpu` WORKER PROCESSES, worker i advancing the chains
        i, i + ncpu, i + 2 ncpu, ... (the reference's fan-out: mc3/chain.py:163,190-191) and
        meeting the others once per refresh (ls_run_worker).  Needs shared=True.  Returns the
        wall seconds from the moment all workers stand ready to the moment all are done
        (process start-up is not the chains' time: the reference's hub starts its workers
        once per run)."""
        import os
        import signal
        import time
        if self._barrier is None:
            raise RuntimeError('run_processes needs LockstepRun(..., shared=True)')
        # what ls_run_worker would refuse -- BEFORE anybody is forked: a worker that returns
        # without attending the first meeting leaves the others (and the parent) waiting
        ncpu, nrefresh = int(ncpu), int(nrefresh)
        if ncpu < 1 or nrefresh < 0:
            raise ValueError(f'run_processes: ncpu = {ncpu}, nrefresh = {nrefresh}')
        if self.M0 + (int(self.S.refresh_done) + nrefresh)*self.nchains > self.zlen:
            raise ValueError('run_processes: Z has no room for that many refreshes')
        self.L.ls_run_worker.restype = ctypes.c_int
        # two barriers: [0] the workers' meeting per refresh, [1] start / end with the parent
        # (the fourth word of each: "called off", ls_barrier_abandon)
        self._barrier[:] = (0, 0, ncpu, 0, 0, 0, ncpu + 1, 0)
        bar = (ctypes.c_int32*8).from_buffer(self._barrier)
        pids = []
        for wid in range(ncpu):
            pid = os.fork()
            if pid == 0:
                rc = 1
                try:
                    if os.environ.get('ORACLE_TEST_WORKER_DIES') == str(wid):
                        os._exit(7)                    # (test hook: this worker never shows up)
                    rc = self.L.ls_run_worker(
                        ctypes.byref(self.P), ctypes.byref(self.S), ctypes.c_int64(nrefresh),
                        ctypes.c_int(wid), ctypes.c_int(ncpu), bar)
                finally:
                    os._exit(0 if rc == 0 else 3)
            pids.append(pid)
        workers = ctypes.cast(ctypes.addressof(bar), ctypes.POINTER(ctypes.c_int32))
        start_end = ctypes.cast(ctypes.addressof(bar) + 16, ctypes.POINTER(ctypes.c_int32))
        sense = ctypes.c_int(0)
        status = {}

        def attend():
            """The parent's half of a start / end meeting, looking after the children while it
            waits: one that is gone for another reason than success can never show up, so the
            run is called off, the others are ended, and the caller hears of it."""
            if self.L.ls_barrier_arrive(start_end, ctypes.byref(sense)):
                return
            while True:
                for _ in range(2000):
                    if self.L.ls_barrier_passed(start_end, sense) != 0:
                        return
                for p in pids:
                    if p not in status:
                        done, st = os.waitpid(p, os.WNOHANG)
                        if done:
                            status[p] = st
                if any(st != 0 for st in status.values()):
                    self.L.ls_barrier_abandon(workers)
                    self.L.ls_barrier_abandon(start_end)
                    deadline = time.time() + 5.0
                    for p in pids:
                        while p not in status:
                            done, st = os.waitpid(p, os.WNOHANG)
                            if done:
                                status[p] = st
                            elif time.time() > deadline:
                                os.kill(p, signal.SIGKILL)
                                status[p] = os.waitpid(p, 0)[1]
                            else:
                                time.sleep(0.01)
                    raise RuntimeError(f'worker processes failed: '
                                       f'{[status[p] for p in pids]}')
                time.sleep(0.0005)

        attend()
        t0 = time.perf_counter()
        attend()
        dt = time.perf_counter() - t0
        bad = [status[p] if p in status else os.waitpid(p, 0)[1] for p in pids]
        if any(bad):
            raise RuntimeError(f'worker processes failed: {bad}')
        self.S.refresh_done += nrefresh
        return dt

    def run_tape(self, kind, value, fma_dot=True, trace=True):
        """The reference's own sequential loop (ls_run_tape) fed a tape of its
        np.random draws.  Returns a dict: rc (0 = the whole tape was consumed with
        every draw of the recorded kind), numaccept, best_chisq, bestx, steps and
        -- with trace -- per-step chisq / flags in visiting order."""
        kind = np.ascontiguousarray(kind, dtype=np.int8)
        value = _f(value)
        nmax = (self.nzchain + 2)*self.thinning*self.nchains
        if trace:
            self.trace_chisq = np.full(nmax, np.nan)
            self.trace_flag = np.zeros(nmax, np.int32)
            self.S.trace_chisq = self.trace_chisq.ctypes.data_as(_dp)
            self.S.trace_flag = self.trace_flag.ctypes.data_as(_ip)
        else:
            self.S.trace_chisq = None
            self.S.trace_flag = None
        ibest = int(np.argmax(self.log_post[0:self.M0]))      # mcmc_driver.py:273-278
        gbest = ctypes.c_double(-2.0*self.log_post[ibest])
        gbestx = self.Z[ibest].copy()
        nacc = ctypes.c_int64(0)
        steps = ctypes.c_int64(0)
        self.L.ls_run_tape.restype = ctypes.c_int
        rc = self.L.ls_run_tape(
            ctypes.byref(self.P), ctypes.byref(self.S), value.ctypes.data_as(_dp),
            kind.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)),
            ctypes.c_int64(len(value)), ctypes.c_int(int(fma_dot)), ctypes.byref(nacc),
            ctypes.byref(gbest), gbestx.ctypes.data_as(_dp), ctypes.byref(steps))
        n = int(steps.value)
        return {'rc': int(rc), 'numaccept': int(nacc.value), 'best_chisq': gbest.value,
                'bestx': gbestx, 'steps': n,
                'trace_chisq': self.trace_chisq[:n] if trace else None,
                'trace_flag': self.trace_flag[:n] if trace else None}


def philox4x32_10(ctr, key):
    """Philox through the C restatement (for the known-answer test)."""
    L = ok.lib()
    c = (ctypes.c_uint32*4)(*ctr)
    k = (ctypes.c_uint32*2)(*key)
    o = (ctypes.c_uint32*4)()
    L.ls_philox4x32_10(c, k, o)
    return tuple(int(v) for v in o)
