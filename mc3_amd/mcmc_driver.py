"""mcmc(): the hub of an MCMC run, with the chains on the GPU.

Same 32 positional arguments and same output dictionary as the reference's
``mc3.mcmc_driver.mcmc`` (mc3/mcmc_driver.py:18-378), which ``sample()`` calls
at mc3/sampler_driver.py:448-457.  What changes is where the work happens: the
reference forks ``ncpu`` Python workers that share memory and round-trips two
pipe messages per worker per DEMC iteration (:143-221,300-307); here the walker
ensemble lives in device memory (mc3_amd.engine.DeviceChains) and the hub only
seeds the initial population, asks for blocks of refreshes and reports.
"""
import random
import time

import numpy as np

from . import models
from . import parallel
from . import stats as ms


class _Value:
    def __init__(self, value):
        self.value = value


class ChainView:
    """What update_output / calc_bestfit_statistics read from a reference Chain
    (mc3/stats/stats.py:813-871), backed by arrays copied from the device."""

    def __init__(self, engine, func, args, kwargs, data, wlike, offset):
        self._engine = engine
        self.func, self.args, self.kwargs = func, args, kwargs
        self.data = data
        self.wlike = wlike
        self._offset = offset
        self.thinning = engine.thinning
        self.prior, self.priorlow, self.priorup = (engine.prior, engine.priorlow,
                                                  engine.priorup)
        self.pstep = engine.pstep
        self.ifree = engine.ifree
        self.nfree = engine.nfree

    def eval_model(self, params, ret='model'):
        """Chain.eval_model (mc3/chain.py:302-340): the model through the user's
        callable, chisq through the device."""
        params = np.asarray(params, float)
        if self.wlike:
            model = self.func(params[0:-3], *self.args, **self.kwargs)
        else:
            model = self.func(params, *self.args, **self.kwargs)
        if ret == 'model':
            return model
        if np.any(model == np.inf):
            chisq = np.inf
        else:
            chisq = float(self._engine.eval_chisq(params))
        return [model, chisq] if ret == 'both' else chisq


def mcmc(data, uncert, func, params, indparams, indparams_dict,
         pmin, pmax, pstep, prior, priorlow, priorup,
         nchains, ncpu, nsamples, sampler,
         wlike, fit_output, grtest, grbreak, grnmin, burnin, thinning,
         fgamma, fepsilon, hsize, kickoff, savefile, resume, log,
         pnames, texnames,
         seed=None, device=0, comm=None, engine_factory=None):
    """Run the chains and assemble the output dictionary (see module docstring).

    ``ncpu`` is accepted for signature compatibility and ignored: the fan-out
    over worker processes (mcmc_driver.py:204-221) is replaced by the device.
    Extra keywords: ``seed`` (Philox key; default random.randint(0, 100000) like
    chain.py:180), ``device``, ``comm`` (mc3_amd.parallel communicator) and
    ``engine_factory`` (tests inject a CPU checker; default DeviceChains).
    """
    data = np.asarray(data, float)
    params = np.array(params, float)
    pstep = np.asarray(pstep, float)
    pmin = np.asarray(pmin, float)
    pmax = np.asarray(pmax, float)
    nfree = int(np.sum(pstep > 0))
    ifree = np.where(pstep > 0)[0]
    ishare = np.where(pstep < 0)[0]
    comm = comm or parallel.SingleRank()
    if engine_factory is None:
        from .engine import DeviceChains as engine_factory

    if resume:
        oldrun = np.load(savefile)
        zold = oldrun['posterior']
        zchain_old = oldrun['zchain']
        pre_zsize = M0 = np.shape(zold)[0]
    else:
        pre_zsize = M0 = hsize*nchains

    # sizes: mcmc_driver.py:129-141
    nzchain = int(np.ceil(nsamples/nchains/thinning))
    niter = nzchain*thinning
    zlen = pre_zsize + nzchain*nchains
    burnin = int(burnin)
    if not resume and niter < burnin:
        log.error(
            f"The number of burned-in samples ({burnin}) is greater than "
            f"the number of iterations per chain ({niter})")
    zburn = int(burnin/thinning)

    # GR minimum sample size: mcmc_driver.py:186-198
    if grnmin >= 1:
        grnmin = int(grnmin/thinning)
    elif grnmin > 0:
        grnmin = int(grnmin*nchains*(nzchain - zburn))
    elif grnmin < 0:
        log.error(
            "Invalid 'grnmin' argument (minimum number of samples to "
            "stop the MCMC under GR convergence), must either be grnmin > 1"
            "to set the minimum number of samples, or 0 < grnmin < 1"
            "to set the fraction of samples required to evaluate.")
    grnmin += int(M0 + zburn*nchains)

    lowered = models.lower(func, params, indparams, indparams_dict, len(data),
                           wlike, pstep)
    data_dev = data if lowered.offset is None else data - lowered.offset
    if seed is None:
        seed = comm.broadcast(random.randint(0, 100000))
    chain0, nlocal = comm.partition(nchains)
    engine = engine_factory(
        data=data_dev, uncert=uncert, indep=lowered.indep, params=params,
        pstep=pstep, pmin=pmin, pmax=pmax, prior=prior, priorlow=priorlow,
        priorup=priorup, nchains=nchains, sampler=sampler, wlike=wlike,
        thinning=thinning, fgamma=fgamma, fepsilon=fepsilon, hsize=hsize,
        model=lowered.family, nmodel=lowered.ncoef, seed=seed, device=device,
        chain0=chain0, nlocal=nlocal, M0=M0, zlen=zlen,
        **({'p2p': True} if comm.wants_p2p else {}))
    view = ChainView(engine, func, indparams, indparams_dict, data, wlike,
                     lowered.offset)
    bestp = np.copy(params)
    best_log_post = np.inf
    numaccept0 = 0
    chainsize0 = np.tile(hsize, nchains)

    if resume:
        # mcmc_driver.py:178-184,223-227; chain.py:166-169
        Z0 = np.array(zold)
        lp0 = np.array(oldrun['log_post'])
        last = np.array([np.where(zchain_old == c)[0][-1] for c in range(nchains)])
        engine.set_population(Z0, lp0, X0=Z0[last], chisq0=-2*lp0[last])
        numaccept0 = int(oldrun['acceptance_rate']/100.0*pre_zsize)
        bestp[:] = oldrun['bestp']
        best_log_post = float(oldrun['best_log_post'])
        chainsize0 = np.array([np.sum(zchain_old == c) for c in range(nchains)])
        zchain0 = np.array(zchain_old, np.int32)
    else:
        # Initial population: M0 in-bounds draws with finite chisq, at most
        # 100*M0 failures (mcmc_driver.py:229-270).  Trials are drawn from the
        # global NumPy stream exactly like the reference, in blocks; the device
        # evaluates each block's chisq in one launch.
        Z0 = lp0 = None
        if comm.rank == 0:
            x0, sigma = params[ifree], pstep[ifree]
            lo, hi = pmin[ifree], pmax[ifree]
            Z0 = np.empty((M0, nfree))
            lp0 = np.empty(M0)
            stale = np.array(params, float)
            filled = failed = 0
            while filled < M0 and failed < 100*M0:
                block = max(256, 2*(M0 - filled))
                if kickoff == 'normal':
                    trial = np.random.normal(x0, sigma, (block, nfree))
                elif kickoff == 'uniform':
                    trial = np.random.uniform(lo, hi, (block, nfree))
                else:
                    log.error(f"Invalid kickoff '{kickoff}'")
                # Bounds are tested on the full vector BEFORE the shared parameters
                # are refreshed, i.e. with the copies left by the last in-bounds trial
                # (mcmc_driver.py:249-255) -- a sequential rule, kept on the host; only
                # the chisq evaluations are batched on the device.
                full = np.empty((block, len(params)))
                inside = np.zeros(block, bool)
                for i in range(block):
                    stale[ifree] = trial[i]
                    if np.any(stale > pmax) or np.any(stale < pmin):
                        continue
                    for s in ishare:
                        stale[s] = stale[-int(pstep[s]) - 1]
                    inside[i] = True
                    full[i] = stale
                lp = np.full(block, -np.inf)
                if np.any(inside):
                    with np.errstate(all='ignore'):
                        lp[inside] = -0.5*engine.eval_chisq(full[inside])
                good = inside & np.isfinite(lp)
                for i in range(block):
                    if filled == M0 or failed == 100*M0:
                        break
                    if good[i]:
                        Z0[filled] = trial[i]
                        lp0[filled] = lp[i]
                        filled += 1
                    else:
                        failed += 1
            if filled < M0 - 1:
                Z0 = None
        Z0, lp0 = comm.broadcast((Z0, lp0))
        if Z0 is None:
            log.error(
                'Cannot populate an initial sample set of parameters, try '
                'updating the parameters initial guess to avoid sampling '
                'beyond the parameter boundaries or where the model returns '
                'non-finite values.')
        engine.set_population(Z0, lp0)
        izbest = np.argmax(lp0)
        best_log_post = lp0[izbest]
        bestp[ifree] = np.copy(Z0[izbest])
        for s in ishare:
            bestp[s] = bestp[-int(pstep[s]) - 1]
        if fit_output is not None:
            bestp[:] = np.copy(fit_output['bestp'])
            best_log_post = fit_output['best_log_post']
        zchain0 = -np.ones(M0, np.int32)
    comm.setup(engine)
    best0 = (np.copy(bestp), float(best_log_post))

    output = {
        'pnames': pnames,
        'texnames': texnames,
        'pstep': pstep,
        'ifree': ifree,
        'burnin': zburn,
    }

    def snapshot(final=False):
        """Pull state (and rows) from the device into the Chain-shaped view."""
        if final:
            comm.finish(engine)
        else:
            comm.quiesce(engine)
        st = comm.reduce_state(engine.state())
        done = st['refresh_done']
        # global best: chain.py:268-274 applied to every accepted state in
        # (iteration, chain) order, starting from the initial best
        bp, blp = np.copy(best0[0]), best0[1]
        cand = np.flatnonzero(st['best_iter'] >= 0)
        if len(cand):
            order = cand[np.lexsort((cand, st['best_iter'][cand]))]
            k = order[np.argmin(st['best_chisq'][order])]
            if st['best_chisq'][k] < -2*blp:
                bp[ifree] = st['bestx'][k]
                for s in ishare:
                    bp[s] = bp[-int(pstep[s]) - 1]
                blp = -0.5*st['best_chisq'][k]
        view.bestp = bp
        view.best_log_post = blp
        view.numaccept = _Value(numaccept0 + int(st['numaccept'].sum()))
        view.outbounds = st['outbounds']
        view.chainsize = chainsize0 + done
        nrows = M0 + done*nchains
        view.zsize = nrows
        return nrows

    def pull_rows(nrows):
        Z, lp, zc = engine.rows(0, nrows)
        zc[:M0] = zchain0
        view.Z, view.log_post, view.zchain = Z, lp, zc

    print("Yippee Ki Yay Monte Carlo!")
    log.msg(f"Start MCMC chains  ({time.ctime()})")
    # ten progress reports: mcmc_driver.py:297-348
    nreport = 10
    edges = np.unique(np.ceil(np.arange(1, nreport + 1)*nzchain/nreport).astype(int))
    done = 0
    for edge in edges:
        todo = int(edge) - done
        if comm.size > 1 and not comm.native:
            for _ in range(todo):
                engine.run(1)
                comm.exchange(engine, done)
                done += 1
        else:
            # (not waited for here: snapshot() synchronises -- and, on several GPUs,
            # is where the ranks agree that the segment completed everywhere)
            engine.run(todo, sync=False)
            done = int(edge)
        nrows = snapshot()
        log.progressbar((nrows + 1.0 - pre_zsize)/(nzchain*nchains))
        log.msg(
            f"Out-of-bound Trials:\n{view.outbounds}\n"
            f"Best Parameters: (chisq={-2*view.best_log_post:.4f})\n"
            f"{view.bestp[ifree]}", width=80)
        run_gr = grtest and np.all(view.chainsize > (zburn + hsize))
        # the Gelman-Rubin reductions run on the device when the chains started in
        # this call (regular row layout); Z is then only read back for a savefile
        device_gr = run_gr and not resume and hasattr(engine, 'chain_moments') \
            and (comm.size == 1 or comm.native)
        need_rows = (savefile is not None) or (run_gr and not device_gr)
        if need_rows:
            if comm.native:
                comm.finish(engine)
            pull_rows(nrows)
        if savefile is not None and comm.rank == 0:
            ms.update_output(output, view, hsize)
            np.savez(savefile, **output)
        if run_gr:
            if device_gr:
                mean, var = engine.chain_moments(zburn)
                psrf = ms.gelman_rubin_from_moments(mean, var, done - zburn)
            else:
                psrf = ms.gelman_rubin(view.Z, view.zchain, zburn)
            log.msg(f"Gelman-Rubin statistics for free parameters:\n{psrf}",
                    width=80)
            if np.all(psrf < 1.01):
                log.msg("All parameters converged to within 1% of unity.")
            if grbreak > 0.0 and np.all(psrf < grbreak) and nrows > grnmin:
                log.msg(
                    "\nAll parameters satisfy the GR convergence "
                    f"threshold of {grbreak:g}, stopping the MCMC.")
                break

    nrows = snapshot(final=True)
    pull_rows(nrows)
    posterior = ms.update_output(output, view, hsize)
    engine.close()

    Z = output['posterior']
    nsample = len(Z)*thinning
    nzsample = len(posterior) if posterior is not None else 0
    fmt = len(str(nsample))
    log.msg('\nMCMC Summary:\n-------------')
    log.msg(
        f"Number of evaluated samples:        {nsample:{fmt}d}\n"
        f"Number of parallel chains:          {nchains:{fmt}d}\n"
        f"Average iterations per chain:       {nsample//nchains:{fmt}d}\n"
        f"Burned-in iterations per chain:     {burnin:{fmt}d}\n"
        f"Thinning factor:                    {thinning:{fmt}d}\n"
        f"MCMC sample size (thinned, burned): {nzsample:{fmt}d}\n"
        f"Acceptance rate:   {output['acceptance_rate']:.2f}%\n", indent=2)
    return output
