"""DeviceChains: the walker ensemble on one MI355X.

Host-side counterpart of the reference's ``Chain`` workers plus the shared
memory ``mcmc()`` allocates for them (mc3/chain.py:18-155,
mc3/mcmc_driver.py:143-221).  Everything numeric happens in libmc3hip.so; this
class only marshals NumPy arrays through the C-ABI (include/mc3hip.h) and
mirrors the attributes ``update_output`` reads from a Chain
(mc3/stats/stats.py:813-871).
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import check, f64, dptr, iptr, lptr


class DeviceChains:
    """nchains walkers (or the block [chain0, chain0+nlocal) of them) on one GPU.

    model: 'poly'   -> indep = x[ndata],          model = sum_k p[k] x^k
           'linear' -> indep = A[nmodel, ndata],  model = sum_k A[k] p[k]
    The model coefficients are params[0:nmodel]; with wlike=True the last three
    parameters are the wavelet noise parameters (chain.py:317, _dwt.c:76-79).
    """

    def __init__(self, data, uncert, indep, params, pstep, pmin=None, pmax=None,
                 prior=None, priorlow=None, priorup=None, nchains=7,
                 nsamples=None, sampler='snooker', wlike=False, thinning=1,
                 fgamma=1.0, fepsilon=0.0, hsize=10, model='poly', nmodel=None,
                 seed=0, device=0, chain0=0, nlocal=None, M0=None, zlen=None,
                 trace_iters=0, no_graph=False, no_persist=False, p2p=False, zlag=0, wl_nreal=0,
                 no_courier=False, jit=True):
        self._ctx = None
        self.lib = _lib.load()
        self.params = f64(params).copy()
        npar = len(self.params)
        self.pstep = f64(pstep).copy()
        self.pmin = f64(np.tile(-np.inf, npar) if pmin is None else pmin)
        self.pmax = f64(np.tile(np.inf, npar) if pmax is None else pmax)
        zeros = np.zeros(npar)
        self.prior = f64(zeros if prior is None else prior)
        self.priorlow = f64(zeros if priorlow is None else priorlow)
        self.priorup = f64(zeros if priorup is None else priorup)
        self.data = f64(data)
        self.ndata = len(self.data)
        self.uncert = f64(np.ones(self.ndata) if uncert is None else uncert)
        self.indep = f64(indep)
        if sampler not in _lib.SAMPLERS:
            raise ValueError(f"Invalid sampler '{sampler}', select from "
                             "['mrw', 'demc', 'snooker']")
        self.sampler = sampler
        self.wlike = bool(wlike)
        self.thinning = int(thinning)
        self.nchains = int(nchains)
        self.chain0 = int(chain0)
        self.nlocal = self.nchains - self.chain0 if nlocal is None else int(nlocal)
        self.ifree = np.where(self.pstep > 0)[0]
        self.ishare = np.where(self.pstep < 0)[0]
        self.nfree = len(self.ifree)
        self.hsize = int(hsize)
        if nmodel is None:
            nmodel = npar - 3 if self.wlike else npar
        self.nmodel = int(nmodel)
        self.model = model
        want = (self.ndata,) if model == 'poly' else (self.nmodel, self.ndata)
        if self.indep.shape != want:
            raise ValueError(f"indep has shape {self.indep.shape}, the '{model}' "
                             f"model needs {want}")
        # sizes: mcmc_driver.py:127-134
        self.M0 = self.hsize*self.nchains if M0 is None else int(M0)
        if zlen is None:
            self.nzchain = int(np.ceil(nsamples/self.nchains/self.thinning))
            zlen = self.M0 + self.nzchain*self.nchains
        else:
            self.nzchain = (int(zlen) - self.M0)//self.nchains
        self.zlen = int(zlen)
        self.seed = int(seed)
        # chain.py:175
        self.gamma = float(fgamma*2.38/np.sqrt(2*self.nfree)) if self.nfree else 0.0
        self.fepsilon = float(fepsilon)

        P = _lib.Problem()
        P.struct_bytes = ctypes.sizeof(_lib.Problem)
        P.device = int(device)
        self._device = int(device)
        P.nchains, P.chain0, P.nlocal = self.nchains, self.chain0, self.nlocal
        P.nparams, P.ndata = npar, self.ndata
        P.sampler = _lib.SAMPLERS[sampler]
        P.wlike = int(self.wlike)
        P.model_kind = {'poly': _lib.MODEL_POLY, 'linear': _lib.MODEL_LINEAR}[model]
        P.nmodel = self.nmodel
        P.thinning = self.thinning
        # (run-time specialisation is something the caller promises, include/mc3hip.h
        # MC3HIP_FLAG_JIT: only with a compiler at hand and not switched off -- otherwise the
        # library plans with its own instances)
        from . import jit as _jit
        self._jit = bool(jit) and not no_persist and _jit.available() and \
            not os.environ.get('MC3HIP_NO_JIT')
        base_flags = (_lib.FLAG_NO_GRAPH if no_graph else 0) | \
            (_lib.FLAG_NO_PERSIST if no_persist else 0) | \
            (_lib.FLAG_P2P if p2p else 0) | (_lib.FLAG_NO_COURIER if no_courier else 0)
        P.flags = base_flags | (_lib.FLAG_JIT if self._jit else 0)
        self.no_courier = bool(no_courier)
        P.trace_iters = int(trace_iters)
        P.M0, P.zlen = self.M0, self.zlen
        P.seed = self.seed
        P.gamma, P.fepsilon = self.gamma, self.fepsilon
        # the declared cadence of a run sharded over GPUs (include/mc3hip.h mc3hip_problem.zlag)
        self.zlag = int(zlag)
        P.zlag = self.zlag
        # a wavelet series padded to 2^M points: how many are data (mc3hip_problem.wl_nreal)
        P.wl_nreal = int(wl_nreal)
        for name in ('params', 'pstep', 'pmin', 'pmax', 'prior', 'priorlow',
                     'priorup', 'indep', 'data', 'uncert'):
            setattr(P, name, dptr(getattr(self, name)))
        ctx = ctypes.c_void_p()
        check(self.lib.mc3hip_create(ctypes.byref(P), ctypes.byref(ctx)))
        self._ctx = ctx
        self.trace_iters = int(trace_iters)
        # run-time specialisation of the one-wave loops for this model, where the library
        # asks for one (mc3_amd/jit.py; MC3HIP_NO_JIT=1 / jit=False: the general chain)
        self.jit_served = ''
        if self._jit:
            self.jit_served = _jit.specialise(self)
            if not self.jit_served and self.jit_request():
                # the promise could not be kept (the compiler failed, or the audit refused the
                # code object): the context was planned FOR the specialised loop -- make it
                # again without the promise, so that the library's own plan runs the shape
                self.lib.mc3hip_destroy(self._ctx)
                self._ctx = None
                self._jit = False
                P.flags = base_flags
                ctx = ctypes.c_void_p()
                check(self.lib.mc3hip_create(ctypes.byref(P), ctypes.byref(ctx)))
                self._ctx = ctx

    @property
    def plan(self):
        """How the library runs this problem right now (summation order, launch
        structure); the launch structure can change if a persistent launch had to
        be redone (see mc3hip_sync)."""
        plan = _lib.Plan()
        check(self.lib.mc3hip_get_plan(self._ctx, ctypes.byref(plan)))
        return {'vlanes': plan.vlanes, 'tile': plan.tile,
                'waves_per_chain': plan.waves_per_chain,
                'graph_nodes': plan.graph_nodes,
                'persistent': plan.persistent, 'fast': plan.fast,
                'comm_graph': plan.comm_graph, 'recoveries': plan.recoveries,
                'tile_tree': plan.tile_tree, 'resident': plan.resident,
                'chains_per_workgroup': plan.chains_per_workgroup, 'jit': plan.jit,
                'capacity': plan.capacity,
                'zlag': plan.zlag, 'remote_delay_ns': plan.remote_delay_ns,
                'owner_waves': plan.owner_waves}

    @property
    def last_kernel(self):
        """Device name of the chain kernel this context launched last ('' before the first
        launch; 'jit:<kernel>' for a loop specialised at run time)."""
        buf = ctypes.create_string_buffer(1024)
        check(self.lib.mc3hip_debug_last_kernel(self._ctx, buf, len(buf)))
        return buf.value.decode().partition(' ')[2]

    def jit_request(self):
        """The loop instantiation the library would like compiled for this model ('' = none)."""
        buf = ctypes.create_string_buffer(256)
        check(self.lib.mc3hip_jit_request(self._ctx, buf, len(buf)))
        return buf.value.decode()

    def jit_install(self, image, name):
        keep = ctypes.create_string_buffer(image, len(image))
        check(self.lib.mc3hip_jit_install(self._ctx, ctypes.cast(keep, ctypes.c_void_p),
                                          len(image), name.encode()))

    # -- lifetime --------------------------------------------------------------
    def close(self):
        if self._ctx is not None:
            self.lib.mc3hip_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- the hot path ----------------------------------------------------------
    def init_population(self, kickoff='normal', seed=None):
        """The initial population on the device (mc3/mcmc_driver.py:229-270): returns
        (Z0 [M0, nfree], log_post0 [M0], filled, failed); rows beyond ``filled`` are
        zeros.  Draws: Philox(seed) on a stream of its own."""
        if kickoff not in ('normal', 'uniform'):
            raise ValueError(f"Invalid kickoff '{kickoff}'")
        Z0 = np.zeros((self.M0, self.nfree))
        lp0 = np.zeros(self.M0)
        filled, failed = ctypes.c_int64(), ctypes.c_int64()
        check(self.lib.mc3hip_init_population(
            self._ctx, int(kickoff == 'uniform'), int(self.seed if seed is None else seed),
            dptr(Z0), dptr(lp0), ctypes.byref(filled), ctypes.byref(failed)))
        return Z0, lp0, int(filled.value), int(failed.value)

    def set_population(self, Z0, log_post0, X0=None, chisq0=None):
        Z0 = f64(Z0)
        log_post0 = f64(log_post0)
        if Z0.shape != (self.M0, self.nfree) or log_post0.shape != (self.M0,):
            raise ValueError('initial population must be [M0, nfree] and [M0]')
        X0 = None if X0 is None else f64(X0)
        chisq0 = None if chisq0 is None else f64(chisq0)
        check(self.lib.mc3hip_set_population(self._ctx, dptr(Z0), dptr(log_post0),
                                             dptr(X0), dptr(chisq0)))

    def run(self, nrefresh=None, sync=True):
        """Advance every local chain by nrefresh*thinning iterations."""
        if nrefresh is None:
            nrefresh = self.nzchain - self.refresh_done
        check(self.lib.mc3hip_run(self._ctx, int(nrefresh)))
        if sync:
            check(self.lib.mc3hip_sync(self._ctx))
        return self

    def sync(self):
        check(self.lib.mc3hip_sync(self._ctx))

    def recycle(self, keep=None, seed=None):
        """Streaming runs (mc3hip_recycle): the last ``keep`` refreshes (default 1 + zlag, what
        the next refresh looks at) become the first ones, the rest of Z is free again, the
        chains carry on under the Philox key ``seed`` (default: the old key + 1)."""
        keep = 1 + self.zlag if keep is None else int(keep)
        self.seed = (self.seed + 1) % (1 << 64) if seed is None else int(seed)
        check(self.lib.mc3hip_recycle(self._ctx, keep, self.seed))

    def last_run_ms(self):
        ms = ctypes.c_float()
        check(self.lib.mc3hip_last_run_ms(self._ctx, ctypes.byref(ms)))
        return float(ms.value)

    # -- read-back ---------------------------------------------------------------
    def rows(self, row0=0, nrows=None):
        """(Z, log_post, zchain) for rows [row0, row0+nrows)."""
        if nrows is None:
            nrows = self.zlen - row0
        Z = np.empty((nrows, self.nfree))
        lp = np.empty(nrows)
        zc = np.empty(nrows, np.int32)
        check(self.lib.mc3hip_get_rows(self._ctx, int(row0), int(nrows), dptr(Z),
                                       dptr(lp), iptr(zc)))
        return Z, lp, zc

    def rows_settled_into(self, row0, Z, lp, zc):
        """Rows [row0, row0 + len(Z)) of refreshes a sync has confirmed, straight into the
        caller's (C-contiguous) arrays, on a stream of their own: a run in flight is neither
        waited for nor disturbed (include/mc3hip.h mc3hip_get_rows_settled; one GPU)."""
        assert Z.flags.c_contiguous and lp.flags.c_contiguous and zc.flags.c_contiguous
        check(self.lib.mc3hip_get_rows_settled(self._ctx, int(row0), len(Z), dptr(Z), dptr(lp),
                                               iptr(zc)))

    def put_rows(self, row0, Z, log_post=None):
        """Install rows another rank produced (host-staged exchange)."""
        Z = f64(Z)
        lp = None if log_post is None else f64(log_post)
        check(self.lib.mc3hip_put_rows(self._ctx, int(row0), len(Z), dptr(Z), dptr(lp)))

    def state(self):
        nl, nf = self.nlocal, self.nfree
        out = {
            'X': np.empty((nl, nf)), 'chisq': np.empty(nl),
            'numaccept': np.empty(nl, np.int64), 'outbounds': np.empty(nf, np.int64),
            'best_chisq': np.empty(nl), 'best_iter': np.empty(nl, np.int64),
            'bestx': np.empty((nl, nf)),
        }
        done = ctypes.c_int64()
        check(self.lib.mc3hip_get_state(
            self._ctx, dptr(out['X']), dptr(out['chisq']), lptr(out['numaccept']),
            lptr(out['outbounds']), dptr(out['best_chisq']), lptr(out['best_iter']),
            dptr(out['bestx']), ctypes.byref(done)))
        out['refresh_done'] = int(done.value)
        return out

    @property
    def refresh_done(self):
        done = ctypes.c_int64()
        check(self.lib.mc3hip_get_state(self._ctx, None, None, None, None, None,
                                        None, None, ctypes.byref(done)))
        return int(done.value)

    def trace(self, iters=None):
        iters = self.trace_iters if iters is None else int(iters)
        chisq = np.empty((iters, self.nlocal))
        flags = np.empty((iters, self.nlocal), np.int32)
        check(self.lib.mc3hip_get_trace(self._ctx, iters, dptr(chisq), iptr(flags)))
        return chisq, flags

    def chain_moments(self, burn=0, nrefresh=None):
        """Per-chain mean and variance of the free parameters over the rows written
        so far (first ``burn`` rows of each chain skipped), computed on the device:
        the inputs of the Gelman-Rubin test (mc3/stats/gelman.py).  ``nrefresh``: over
        that many refreshes' rows as they stand in Z (rows installed with put_rows)."""
        mean = np.empty((self.nchains, self.nfree))
        var = np.empty((self.nchains, self.nfree))
        if nrefresh is None:
            check(self.lib.mc3hip_chain_moments(self._ctx, int(burn), dptr(mean), dptr(var)))
        else:
            check(self.lib.mc3hip_chain_moments_rows(self._ctx, int(nrefresh), int(burn),
                                                     dptr(mean), dptr(var)))
        return mean, var

    def posterior_summary(self, burn=0, ranks=(), nrefresh=None):
        """Mean, population variance, minimum, maximum ([nfree] each) and the order
        statistics at the 0-based ``ranks`` ([nranks, nfree]) of every free parameter
        over the rows of refreshes [burn, done) -- computed where the rows are."""
        ranks = np.ascontiguousarray(ranks, dtype=np.int64)
        nf = self.nfree
        mean, var, vmin, vmax = (np.empty(nf) for _ in range(4))
        order = np.empty((len(ranks), nf))
        check(self.lib.mc3hip_posterior_summary(
            self._ctx, int(nrefresh or 0), int(burn), len(ranks), lptr(ranks), dptr(order),
            dptr(mean), dptr(var), dptr(vmin), dptr(vmax)))
        return {'mean': mean, 'var': var, 'min': vmin, 'max': vmax, 'order': order,
                'n': ((nrefresh or self.refresh_done) - int(burn))*self.nchains}

    def chisq_rows(self, row0=0, nrows=None):
        """-2 (log_post - log_prior) of rows [row0, row0+nrows) (stats.py:821)."""
        if nrows is None:
            nrows = self.zlen - row0
        out = np.empty(nrows)
        check(self.lib.mc3hip_get_chisq(self._ctx, int(row0), int(nrows), dptr(out)))
        return out

    def kde_pdf(self, sample, x, cov):
        return kde_pdf(sample, x, cov, self._device)

    # -- Chain.eval_model (chain.py:302-340) on the device ----------------------
    def eval_chisq(self, params):
        """chisq for one parameter vector or an [n, nparams] batch."""
        p = f64(np.atleast_2d(params))
        if p.shape[1] != len(self.params):
            raise ValueError('parameter vectors must have nparams entries')
        out = np.empty(len(p))
        check(self.lib.mc3hip_eval_chisq(self._ctx, dptr(p), len(p), dptr(out)))
        return out[0] if np.ndim(params) == 1 else out

    def expand_free(self, free):
        """[n, nfree] free-parameter rows -> [n, nparams] full vectors (fixed
        values kept, shared parameters copied, chain.py:246-247)."""
        free = f64(np.atleast_2d(free))
        full = np.tile(self.params, (len(free), 1))
        full[:, self.ifree] = free
        for s in self.ishare:
            full[:, s] = full[:, -int(self.pstep[s]) - 1]
        return full

    def eval_model_values(self, params):
        p = f64(params)
        out = np.empty(self.ndata)
        check(self.lib.mc3hip_eval_model(self._ctx, dptr(p), dptr(out)))
        return out

    # -- multi-GPU ----------------------------------------------------------------
    def comm_init(self, unique_id, rank, nranks):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        check(self.lib.mc3hip_comm_init(self._ctx, buf, int(rank), int(nranks)))

    def p2p_export(self):
        """IPC handles (128 bytes) of this rank's Z / log_post replicas."""
        buf = ctypes.create_string_buffer(128)
        check(self.lib.mc3hip_p2p_export(self._ctx, buf))
        return buf.raw

    def p2p_attach(self, rank, handles):
        """handles: the 128-byte export of every rank, in rank order."""
        blob = ctypes.create_string_buffer(b''.join(handles), 128*len(handles))
        check(self.lib.mc3hip_p2p_attach(self._ctx, int(rank), len(handles), blob))

    def comm_gather_log_post(self):
        check(self.lib.mc3hip_comm_gather_log_post(self._ctx))

    def comm_destroy(self):
        check(self.lib.mc3hip_comm_destroy(self._ctx))

    def comm_info(self):
        """(nranks, rank) as RCCL itself reports them for the attached communicator."""
        n, r = ctypes.c_int32(), ctypes.c_int32()
        check(self.lib.mc3hip_comm_info(self._ctx, ctypes.byref(n), ctypes.byref(r)))
        return int(n.value), int(r.value)

    def p2p_detach(self):
        check(self.lib.mc3hip_p2p_detach(self._ctx))

    def p2p_rollback(self):
        """After a peer-to-peer launch gave up on any rank (and every rank has
        synchronised): back to the checkpoint; returns the refreshes [r0, r1) to redo
        through another exchange."""
        r0, r1 = ctypes.c_int64(), ctypes.c_int64()
        check(self.lib.mc3hip_p2p_rollback(self._ctx, ctypes.byref(r0), ctypes.byref(r1)))
        return int(r0.value), int(r1.value)


def device_memory(device=0):
    """(free, total) bytes of device memory."""
    f, t = ctypes.c_int64(), ctypes.c_int64()
    check(_lib.load().mc3hip_device_memory(int(device), ctypes.byref(f), ctypes.byref(t)))
    return int(f.value), int(t.value)


def comm_unique_id():
    buf = ctypes.create_string_buffer(128)
    check(_lib.load().mc3hip_comm_unique_id(buf))
    return buf.raw


def kde_pdf(sample, x, cov, device=0):
    """Gaussian kernel density of a 1D sample at the points x with kernel variance cov
    (scipy.stats.gaussian_kde.evaluate as cred_region uses it, stats.py:395-467)."""
    sample, x = f64(sample), f64(x)
    out = np.empty(len(x))
    check(_lib.load().mc3hip_kde_pdf(int(device), dptr(sample), len(sample), dptr(x), len(x),
                                     float(cov), dptr(out)))
    return out


def dwt_daub4(array, inverse=False, device=0):
    """ms.dwt_daub4 on the device (src_c/_dwt.c:154-186)."""
    a = f64(array)
    n = len(a)
    nw = 1
    while nw < n:
        nw <<= 1
    out = np.empty(nw)
    check(_lib.load().mc3hip_dwt_daub4(int(device), dptr(a), n, int(bool(inverse)),
                                       dptr(out)))
    return out
