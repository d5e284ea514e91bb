"""A CPU stand-in for mc3_amd.engine.DeviceChains built on the lockstep oracle.

TEST INFRASTRUCTURE: lets the host logic of mc3_amd.mcmc()/sample() and the
multi-rank exchange run without a GPU (tests inject it through
``engine_factory``), and is the bit-exact expectation for the GPU runs of the
same calls.  The product never imports this.
"""
import numpy as np

from oracle import lockstep as ls


class OracleChains:
    def __init__(self, data, uncert, indep, params, pstep, pmin=None, pmax=None,
                 prior=None, priorlow=None, priorup=None, nchains=7,
                 nsamples=None, sampler='snooker', wlike=False, thinning=1,
                 fgamma=1.0, fepsilon=0.0, hsize=10, model='poly', nmodel=None,
                 seed=0, device=0, chain0=0, nlocal=None, M0=None, zlen=None,
                 trace_iters=0, no_graph=False, no_persist=False, p2p=False, nthreads=1,
                 plan=None, zlag=0, no_courier=False, wl_nreal=0):
        if M0 is None:
            M0 = hsize*nchains
        if nsamples is None:
            nsamples = (zlen - M0)*thinning
        self.run_ = ls.LockstepRun(
            data=data, uncert=np.ones(len(data)) if uncert is None else uncert,
            indep=indep, params=params, pstep=pstep, pmin=pmin, pmax=pmax,
            prior=prior, priorlow=priorlow, priorup=priorup, nchains=nchains,
            nsamples=nsamples, sampler=sampler, wlike=wlike, thinning=thinning,
            fgamma=fgamma, fepsilon=fepsilon, hsize=hsize, model=model,
            nmodel=nmodel, seed=seed, chain0=chain0, nlocal=nlocal, M0=M0, zlag=zlag,
            wl_nreal=wl_nreal,
            # (the summation order of the plan the library would choose for this problem:
            # tests that compare with a GPU run pass DeviceChains.plan's entries)
            **({k: plan[k] for k in ('tile', 'vlanes', 'tile_tree') if k in plan}
               if plan else {}))
        r = self.run_
        self.nthreads = nthreads
        self.nchains, self.thinning, self.nfree = r.nchains, r.thinning, r.nfree
        self.chain0 = int(chain0)
        self.nlocal = r.P.nlocal
        self.M0, self.zlen, self.nzchain = r.M0, r.zlen, r.nzchain
        self.params, self.pstep = r.params, r.pstep
        self.prior, self.priorlow, self.priorup = r.prior, r.priorlow, r.priorup
        self.ifree = np.where(r.pstep > 0)[0]
        self.ishare = np.where(r.pstep < 0)[0]
        self.plan = {'vlanes': 1024, 'tile': (plan or {}).get('tile', 0), 'waves_per_chain': 0, 'graph_nodes': 0, 'persistent': 0, 'fast': 0,
                     'comm_graph': 0, 'recoveries': 0, 'tile_tree': (plan or {}).get('tile_tree', 0),
                     'resident': 0}

    def close(self):
        pass

    def init_population(self, kickoff='normal', seed=None):
        return self.run_.init_population(kickoff, seed)

    def set_population(self, Z0, log_post0, X0=None, chisq0=None):
        r = self.run_
        r.set_population(Z0, log_post0)
        if X0 is not None:
            r.X[:] = X0
            r.snap0[:] = X0
            r.chisq[:] = chisq0

    def run(self, nrefresh=None, sync=True):
        self.run_.run(nrefresh=nrefresh, nthreads=self.nthreads)
        return self

    def sync(self):
        pass

    @property
    def refresh_done(self):
        return int(self.run_.S.refresh_done)

    def rows(self, row0=0, nrows=None):
        r = self.run_
        if nrows is None:
            nrows = r.zlen - row0
        s = slice(row0, row0 + nrows)
        return r.Z[s].copy(), r.log_post[s].copy(), r.zchain[s].copy()

    def put_rows(self, row0, Z, log_post=None):
        r = self.run_
        r.Z[row0:row0 + len(Z)] = Z
        if log_post is not None:
            r.log_post[row0:row0 + len(Z)] = log_post
        r.zchain[row0:row0 + len(Z)] = (np.arange(row0, row0 + len(Z)) - r.M0) % r.nchains

    def state(self):
        r = self.run_
        s = slice(self.chain0, self.chain0 + self.nlocal)
        return {'X': r.X[s].copy(), 'chisq': r.chisq[s].copy(),
                'numaccept': r.numaccept[s].copy(), 'outbounds': r.outbounds.copy(),
                'best_chisq': r.best_chisq[s].copy(), 'best_iter': r.best_iter[s].copy(),
                'bestx': r.bestx[s].copy(), 'refresh_done': self.refresh_done}

    def eval_chisq(self, params):
        out = self.run_.eval_chisq(params)
        return out[0] if np.ndim(params) == 1 else out

    def eval_model_values(self, params):
        return self.run_.eval_model(params)

    def expand_free(self, free):
        free = np.atleast_2d(np.asarray(free, float))
        full = np.tile(self.params, (len(free), 1))
        full[:, self.ifree] = free
        for s in self.ishare:
            full[:, s] = full[:, -int(self.pstep[s]) - 1]
        return full
