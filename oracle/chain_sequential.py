"""oracle/chain_sequential.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference's chain update in its own SEQUENTIAL form: one worker process
(ncpu=1) that owns every chain and visits them one after another inside each
iteration, so chain j+1 already sees chain j's update (mc3/chain.py:190).
Random numbers come from NumPy's legacy MT19937 ``RandomState`` in exactly the
reference's call order, so with the two seeds fixed this
reproduces a deterministic reference run bit for bit:

  hub_seed    what ``np.random.seed`` was set to before ``mc3.sample``/``mcmc``
              (drives the initial population, mcmc_driver.py:229-262);
  chain_seed  the value ``random.randint(0, 100000)`` returned inside the
              worker (chain.py:180; stubbed to a constant when the goldens were
              made).

Pinned by tests/test_oracle_golden.py against tests/golden/seq_*.npz, which
tests/golden/make_golden.py generated from the reference itself.
"""
import numpy as np

from . import kernels as ok


def _default_chisq(model, data, uncert, params, prior, priorlow, priorup, wlike):
    if wlike:
        return ok.dwt_chisq(model, data, params, prior, priorlow, priorup)
    return ok.chisq(model, data, uncert, params, prior, priorlow, priorup)


def ref_chisq_fn():
    """chisq through the reference's own compiled C (oracle/_ref), with the
    Python-side prior masking of stats.py:211-216 / 278-284."""
    mods = ok.load_ref()
    if mods is None:
        return None
    cs, dwt = mods

    def fn(model, data, uncert, params, prior, priorlow, priorup, wlike):
        iprior = (priorlow > 0) & (priorup > 0)
        dprior = (params - prior)[iprior]
        if wlike:
            return dwt.chisq(params, model, data, dprior, priorlow[iprior],
                             priorup[iprior])
        return cs.chisq(model, data, uncert, dprior, priorlow[iprior],
                        priorup[iprior])
    return fn


class SequentialRun:
    """State of one reference-style run: everything ``Chain`` and ``mcmc``
    keep in shared memory (mcmc_driver.py:143-202)."""

    def __init__(self, data, uncert, func, params, indparams=(), pmin=None,
                 pmax=None, pstep=None, prior=None, priorlow=None, priorup=None,
                 nchains=7, nsamples=None, sampler='snooker', wlike=False,
                 thinning=1, fgamma=1.0, fepsilon=0.0, hsize=10,
                 kickoff='normal', chisq_fn=None):
        self.data = np.asarray(data, float)
        self.uncert = np.asarray(uncert, float)
        self.func = func
        self.args = tuple(indparams)
        self.params = np.array(params, float)
        npar = len(self.params)
        self.pmin = np.tile(-np.inf, npar) if pmin is None else np.asarray(pmin, float)
        self.pmax = np.tile(np.inf, npar) if pmax is None else np.asarray(pmax, float)
        self.pstep = np.asarray(pstep, float)
        zeros = np.zeros(npar)
        self.prior = zeros if prior is None else np.asarray(prior, float)
        self.priorlow = zeros if priorlow is None else np.asarray(priorlow, float)
        self.priorup = zeros if priorup is None else np.asarray(priorup, float)
        self.nchains = int(nchains)
        self.sampler = sampler
        self.wlike = bool(wlike)
        self.thinning = int(thinning)
        self.fgamma = fgamma
        self.fepsilon = fepsilon
        self.hsize = int(hsize)
        self.kickoff = kickoff
        self.chisq_fn = chisq_fn or _default_chisq

        self.ifree = np.where(self.pstep > 0)[0]
        self.ishare = np.where(self.pstep < 0)[0]
        self.nfree = len(self.ifree)
        # sizes: mcmc_driver.py:127-134
        self.M0 = self.hsize*self.nchains
        self.nzchain = int(np.ceil(nsamples/self.nchains/self.thinning))
        self.zlen = self.M0 + self.nzchain*self.nchains
        self.Z = np.zeros((self.zlen, self.nfree))
        self.log_post = np.zeros(self.zlen)
        self.zchain = -np.ones(self.zlen, np.int32)
        self.zsize = self.M0
        self.chainsize = np.tile(self.hsize, self.nchains).astype(np.int32)
        self.freepars = np.zeros((self.nchains, self.nfree))
        self.numaccept = 0
        self.outbounds = np.zeros(self.nfree, np.int64)
        self.bestp = np.copy(self.params)
        self.best_log_post = np.inf
        self.trace = []

    # mc3/chain.py:302-340
    def eval_chisq(self, params):
        if self.wlike:
            model = self.func(params[0:-3], *self.args)
        else:
            model = self.func(params, *self.args)
        if np.any(model == np.inf):
            return np.inf
        return self.chisq_fn(model, self.data, self.uncert, params,
                             self.prior, self.priorlow, self.priorup, self.wlike)

    # mc3/mcmc_driver.py:229-278
    def init_population(self, rng):
        x0 = np.copy(self.params[self.ifree])
        sigma = np.copy(self.pstep[self.ifree])
        lo = np.copy(self.pmin[self.ifree])
        hi = np.copy(self.pmax[self.ifree])
        values = np.copy(self.params)
        filled = 0
        failed = 0
        while filled < self.M0 and failed < 100*self.M0:
            if self.kickoff == 'normal':
                trial = rng.normal(x0, sigma)
            else:
                trial = rng.uniform(lo, hi)
            values[self.ifree] = trial
            if np.any(values > self.pmax) or np.any(values < self.pmin):
                failed += 1
                continue
            for s in self.ishare:
                values[s] = values[-int(self.pstep[s]) - 1]
            lp = -0.5*self.eval_chisq(values)
            if not np.isfinite(lp):
                failed += 1
                continue
            self.Z[filled] = values[self.ifree]
            self.log_post[filled] = lp
            filled += 1
        if filled < self.M0 - 1:
            raise ValueError('Cannot populate an initial sample set of parameters')
        ibest = np.argmax(self.log_post[0:self.M0])
        self.best_log_post = self.log_post[ibest]
        self.bestp[self.ifree] = np.copy(self.Z[ibest])
        # NB: the reference's generator draws one more trial than it uses
        # before noticing i == M0 (mcmc_driver.py:245-247): it yields first,
        # tests after.  Consume it so the stream stays aligned.
        if self.kickoff == 'normal':
            rng.normal(x0, sigma)
        else:
            rng.uniform(lo, hi)

    # mc3/chain.py:158-299 with ncpu = 1
    def run_chains(self, rng, record=False):
        nchains, nfree = self.nchains, self.nfree
        ifree, ishare = self.ifree, self.ishare
        index = self.M0 + np.arange(nchains)
        for c in range(nchains):
            self.freepars[c] = np.copy(self.Z[c])
        chisq = -2*self.log_post[np.arange(nchains)]
        nextp = np.copy(self.params)
        gamma = self.fgamma*2.38/np.sqrt(2*nfree)
        njump = 0
        Z, freepars = self.Z, self.freepars
        while True:
            njump += 1
            normal = rng.normal(0, self.pstep[ifree], nfree)
            for c in range(nchains):
                mrfactor = 1.0
                sjump = False
                if self.sampler == 'snooker':
                    iR1 = rng.randint(0, self.zsize)
                    iR2 = rng.randint(1, self.zsize)
                    if iR2 == iR1:
                        iR2 = 0
                    sjump = rng.uniform() < 0.1
                    if sjump:
                        iz = rng.randint(self.zsize)
                        z = Z[iz]
                        if np.all(z == freepars[c]):
                            jump = rng.uniform(1.2, 2.2)*(Z[iR2] - Z[iR1])
                        else:
                            dz = freepars[c] - z
                            zp1 = np.dot(Z[iR1], dz)
                            zp2 = np.dot(Z[iR2], dz)
                            jump = rng.uniform(1.2, 2.2)*(zp1 - zp2)*dz/np.dot(dz, dz)
                    else:
                        jump = gamma*(Z[iR1] - Z[iR2]) + self.fepsilon*normal
                elif self.sampler == 'mrw':
                    jump = normal
                else:
                    r1 = rng.randint(1, nchains)
                    if r1 == c:
                        r1 = 0
                    r2 = (r1 + rng.randint(2, nchains)) % nchains
                    if r2 == c:
                        r2 = (r1 + 1) % nchains
                    jump = gamma*(freepars[r1] - freepars[r2]) + self.fepsilon*normal

                nextp[ifree] = np.copy(freepars[c]) + jump
                outpars = ((nextp < self.pmin) | (nextp > self.pmax))[ifree]
                accepted = False
                nextchisq = np.nan
                if np.any(outpars):
                    self.outbounds += outpars
                else:
                    for s in ishare:
                        nextp[s] = nextp[-int(self.pstep[s]) - 1]
                    nextchisq = self.eval_chisq(nextp)
                    if sjump:
                        cnorm = np.dot(freepars[c] - z, freepars[c] - z)
                        nnorm = np.dot(nextp[ifree] - z, nextp[ifree] - z)
                        mrfactor = (nnorm/cnorm)**(0.5*(nfree - 1))
                    with np.errstate(all='ignore'):
                        accepted = bool(
                            np.exp(0.5*(chisq[c] - nextchisq))*mrfactor
                            > rng.uniform())
                    if accepted:
                        freepars[c] = np.copy(nextp[ifree])
                        chisq[c] = nextchisq
                        self.numaccept += 1
                        if chisq[c] < -2*self.best_log_post:
                            self.bestp[ifree] = freepars[c]
                            for s in ishare:
                                self.bestp[s] = self.bestp[-int(self.pstep[s]) - 1]
                            self.best_log_post = -0.5*chisq[c]
                if record:
                    self.trace.append((c, bool(not np.any(outpars)), nextchisq,
                                       accepted))
                if njump == self.thinning:
                    if self.zsize == self.zlen:
                        return
                    if self.sampler == 'snooker':
                        index[c] = self.zsize
                    self.zsize += 1
                    self.zchain[index[c]] = c
                    Z[index[c]] = np.copy(freepars[c])
                    self.log_post[index[c]] = -0.5*chisq[c]
                    index[c] += nchains
                    self.chainsize[c] += 1
            if njump == self.thinning:
                njump = 0
            if self.sampler in ('mrw', 'demc') and \
                    self.chainsize[0] == self.zlen // nchains:
                return


def run_sequential(hub_seed, chain_seed, record=False, **kw):
    """Initial population under ``hub_seed`` then the chain loop under
    ``chain_seed``; returns the finished SequentialRun."""
    run = SequentialRun(**kw)
    rng = np.random.RandomState(hub_seed)
    with np.errstate(all='ignore'):
        run.init_population(rng)
        rng.seed(chain_seed)
        run.run_chains(rng, record=record)
    return run
