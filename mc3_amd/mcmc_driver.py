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


class _Checkpointer:
    """The periodic save of the hub (``np.savez(savefile, **output)`` at every report,
    mcmc_driver.py:321-324) as a file that GROWS: the per-row arrays are appended in place
    (mc3_amd/npz_stream.py), only the small arrays are rewritten, so a report costs the rows
    that are new, not the run so far; on a worker thread, one write in flight: the chains run
    on while the disk is busy.  Safe without copies: the big arrays of ``output`` are views
    of host rows that were fetched once and are never rewritten (later reports only append
    behind them)."""
    GROWN = ('posterior', 'zchain', 'chisq', 'log_post', 'zmask')

    def __init__(self):
        self.thread = None
        self.error = None
        self.writer = None

    def _writer(self, path, output, max_rows):
        if self.writer is None:
            from .npz_stream import StreamingNpz
            nfree = np.shape(output['posterior'])[1]
            spec = {'posterior': (np.float64, (nfree,), max_rows),
                    'zchain': (np.asarray(output['zchain']).dtype, (), max_rows),
                    'chisq': (np.float64, (), max_rows), 'log_post': (np.float64, (), max_rows),
                    'zmask': (np.intp, (), max_rows)}
            self.writer = StreamingNpz(path, spec)
        return self.writer

    def save(self, path, output, max_rows):
        import threading
        self.wait()
        frozen = dict(output)
        writer = self._writer(path, frozen, max_rows)

        def work():
            try:
                writer.write(frozen)
            except Exception as err:           # noqa: BLE001 -- re-raised by wait()
                self.error = err
        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def finish(self, path, output, max_rows):
        """The complete output dictionary (what sample() saves at the end): the remaining
        rows, every small array, and the file is closed."""
        self.wait()
        self._writer(path, output, max_rows).write(output)
        self.close()

    def close(self):
        if self.thread is not None:
            self.thread.join()
            self.thread = None
        if self.writer is not None:
            self.writer.close()
            self.writer = None

    def wait(self):
        if self.thread is not None:
            self.thread.join()
            self.thread = None
        if self.error is not None:
            err, self.error = self.error, None
            raise err


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


def numpy_stream_population(engine, params, pstep, pmin, pmax, M0, kickoff):
    """The reference's kick-off on its own random stream (mcmc_driver.py:229-270): trial
    vectors from the global NumPy generator in blocks, chisq of a block in one device
    launch, the acceptance rule applied to whole blocks at once.  Returns (Z0, log_post0,
    filled); rows beyond ``filled`` are zero."""
    ifree = np.flatnonzero(pstep > 0)
    ishare = np.flatnonzero(pstep < 0)
    src = np.array([-int(pstep[s]) - 1 for s in ishare], int)
    nfree, npar = len(ifree), len(params)
    Z0 = np.zeros((M0, nfree))
    lp0 = np.zeros(M0)
    values = np.array(params, float)            # (its shared entries: the "stale" copies)
    filled = failed = 0
    nmax = 100*M0
    # The reference tests the boundaries on the FULL vector (fixed entries too) BEFORE
    # refreshing the shared parameters, i.e. with the copies the last trial that passed the
    # test left (:249-255), and refreshes them one after another in index order
    # (chain.py:246-247).  Whole blocks can be judged at once when none of that can matter:
    # the copies are values their sources took inside the sources' boundaries, so the stale
    # test can only bite when a shared parameter's own interval is tighter than its source's
    # (or excludes its initial value), and the order only when a shared parameter's source is
    # itself shared.  Otherwise: trial by trial, the reference's loop as it stands.
    sequential = bool(
        np.any(pmin[ishare] > pmin[src]) or np.any(pmax[ishare] < pmax[src])
        or np.any(values[ishare] > pmax[ishare]) or np.any(values[ishare] < pmin[ishare])
        or np.any(np.isin(src, ishare)))
    ifixed = np.flatnonzero(pstep == 0)
    fixed_outside = bool(np.any(values[ifixed] > pmax[ifixed]) or
                         np.any(values[ifixed] < pmin[ifixed]))
    while filled < M0 and failed < nmax:
        block = max(256, 2*(M0 - filled))
        if kickoff == 'normal':
            trial = np.random.normal(params[ifree], pstep[ifree], (block, nfree))
        else:
            trial = np.random.uniform(pmin[ifree], pmax[ifree], (block, nfree))
        if sequential:
            full = np.empty((block, npar))
            inside = np.zeros(block, bool)
            for i in range(block):
                values[ifree] = trial[i]
                if np.any(values > pmax) or np.any(values < pmin):
                    continue
                for s, q in zip(ishare, src):
                    values[s] = values[q]
                full[i] = values
                inside[i] = True
        else:
            full = np.tile(values, (block, 1))
            full[:, ifree] = trial
            inside = np.all((trial >= pmin[ifree]) & (trial <= pmax[ifree]), axis=1)
            if fixed_outside:
                inside[:] = False                 # (every trial fails the reference's test)
            for s, q in zip(ishare, src):
                full[:, s] = full[:, q]
        lp = np.full(block, -np.inf)
        if np.any(inside):
            with np.errstate(all='ignore'):
                lp[inside] = -0.5*engine.eval_chisq(full[inside])
        good = inside & np.isfinite(lp)
        # in trial order: stop at the M0-th accepted row or at the nmax-th failure
        ngood, nbad = np.cumsum(good), np.cumsum(~good)
        stop = np.flatnonzero((ngood == M0 - filled) & good)
        cut = (stop[0] + 1) if len(stop) else block
        over = np.flatnonzero((nbad == nmax - failed) & ~good)
        if len(over) and over[0] + 1 < cut:
            cut = over[0] + 1
        take = np.flatnonzero(good[:cut])
        Z0[filled:filled + len(take)] = trial[take]
        lp0[filled:filled + len(take)] = lp[take]
        filled += len(take)
        failed += int(np.sum(~good[:cut]))
    return Z0, lp0, filled


def mcmc(data, uncert, func, params, indparams, indparams_dict,
         pmin, pmax, pstep, prior, priorlow, priorup,
         nchains, ncpu, nsamples, sampler,
         wlike, fit_output, grtest, grbreak, grnmin, burnin, thinning,
         fgamma, fepsilon, hsize, kickoff, savefile, resume, log,
         pnames, texnames,
         seed=None, device=0, comm=None, engine_factory=None, kickoff_stream=None,
         zlag=None, wpad=False, _extras=None):
    """Run the chains and assemble the output dictionary (see module docstring).

    ``ncpu`` is accepted for signature compatibility and ignored: the fan-out
    over worker processes (mcmc_driver.py:204-221) is replaced by the device.
    Extra keywords: ``seed`` (Philox key; default random.randint(0, 100000) like
    chain.py:180), ``device``, ``comm`` (mc3_amd.parallel communicator),
    ``engine_factory`` (tests inject a CPU checker; default DeviceChains) and
    ``kickoff_stream``: where the trial vectors of the initial population come from --
    'philox' (default: drawn, tested and evaluated on the device, keyed by ``seed``) or
    'numpy' (the global NumPy stream in the reference's order: with np.random.seed set,
    the population equals the reference's); ``zlag``: the lockstep cadence (include/mc3hip.h
    mc3hip_problem.zlag) -- default 0, the strict lockstep, on ANY number of GPUs: like the
    reference's, the results do not depend on how many workers share the chains; an integer
    or 'auto' (parallel.CADENCE_ZLAG) opts into the relaxed cadence that hides the xGMI hop
    of a sharded run.  It is written to ``output['zlag']``, the savefile and the log, and a
    resumed run must ask for the cadence its savefile was made with; ``wpad``: the wavelet
    likelihood of a series whose length is no power of two -- padded with zero residuals to the
    next one, the zero padding src_c/_dwt.c:82-91 intends (the reference then transforms the
    unpadded length, _dwt.c:94, which only a power of two survives: without the flag such a
    series is refused).
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
    uncert_dev, indep_dev, padded = uncert, lowered.indep, {}
    ndata = len(data)
    if wlike and ndata >= 2 and ndata & (ndata - 1):
        if not wpad:
            raise ValueError(
                f'the wavelet likelihood needs a power-of-two number of data points (got '
                f'{ndata}): the reference zero-pads the residuals and then transforms the '
                'UNPADDED length (src_c/_dwt.c:82-94), which is not defined; wpad=True pads '
                'the series with zero residuals to the next power of two')
        # data 0, uncert 1, the abscissae (design columns) repeat their last value: finite
        # model values there, and residuals that are zero by decree (mc3hip_problem.wl_nreal)
        pad = (1 << int(np.ceil(np.log2(ndata)))) - ndata
        data_dev = np.concatenate([data_dev, np.zeros(pad)])
        uncert_dev = np.concatenate([np.asarray(uncert, float), np.ones(pad)])
        indep_dev = np.concatenate(
            [indep_dev, np.repeat(indep_dev[..., -1:], pad, axis=-1)], axis=-1)
        padded = {'wl_nreal': ndata}
    if seed is None:
        seed = comm.broadcast(random.randint(0, 100000))
    chain0, nlocal = comm.partition(nchains)
    engine_opts = comm.engine_options(device, thinning, zlag)
    zlag_run = int(engine_opts.get('zlag', 0))
    if resume and 'zlag' in oldrun.files and int(oldrun['zlag']) != zlag_run:
        log.error(
            f"The savefile was written by a run in the lockstep cadence zlag = "
            f"{int(oldrun['zlag'])}, this run asks for zlag = {zlag_run}: a resumed run must "
            "continue the same random process (pass the same 'zlag')")
    engine = engine_factory(
        data=data_dev, uncert=uncert_dev, indep=indep_dev, params=params,
        pstep=pstep, pmin=pmin, pmax=pmax, prior=prior, priorlow=priorlow,
        priorup=priorup, nchains=nchains, sampler=sampler, wlike=wlike,
        thinning=thinning, fgamma=fgamma, fepsilon=fepsilon, hsize=hsize,
        model=lowered.family, nmodel=lowered.ncoef, seed=seed, device=device,
        chain0=chain0, nlocal=nlocal, M0=M0, zlen=zlen, **padded, **engine_opts)
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
        # Initial population (mcmc_driver.py:229-270): on the device from Philox(seed)
        # by default; from the global NumPy stream, like the reference, on request or
        # when the engine has no device kick-off.
        Z0 = lp0 = None
        if kickoff not in ('normal', 'uniform'):
            log.error(f"Invalid kickoff '{kickoff}'")
        if kickoff_stream is None:
            kickoff_stream = 'philox' if hasattr(engine, 'init_population') else 'numpy'
        if kickoff_stream == 'philox':
            # (every rank computes the same rows: the draws depend on the seed alone)
            Z0, lp0, filled, _ = engine.init_population(kickoff, seed)
        elif comm.rank == 0:
            Z0, lp0, filled = numpy_stream_population(
                engine, params, pstep, pmin, pmax, M0, kickoff)
        if Z0 is not None and filled < M0 - 1:          # (the reference's test, :264)
            Z0 = None
        if kickoff_stream != 'philox':
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
        # (not in the reference's dictionary: which lockstep cadence produced these rows)
        'zlag': zlag_run,
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

    host = {'rows': 0}
    checkpoint = _Checkpointer()

    # one GPU, a run that started here: the rows of a confirmed segment are copied to the host
    # while the NEXT segment computes (engine.rows_settled_into: a stream of its own)
    overlap = comm.size == 1 and hasattr(engine, 'rows_settled_into')

    def pull_rows(nrows, settled=False):
        """Rows [0, nrows) on the host, fetching only what was not fetched before (one
        host copy for the whole run, filled as the reports come -- or, `settled`, behind
        the launch of the next segment)."""
        if host['rows'] == 0:
            host['Z'] = np.empty((zlen, nfree))
            host['lp'] = np.empty(zlen)
            host['zc'] = np.empty(zlen, np.int32)
        r0 = host['rows']
        if nrows > r0:
            if settled:
                engine.rows_settled_into(r0, host['Z'][r0:nrows], host['lp'][r0:nrows],
                                         host['zc'][r0:nrows])
            else:
                Z, lp, zc = engine.rows(r0, nrows - r0)
                host['Z'][r0:nrows], host['lp'][r0:nrows], host['zc'][r0:nrows] = Z, lp, zc
            if r0 < M0:
                host['zc'][:M0] = zchain0
            host['rows'] = nrows
        view.Z, view.log_post, view.zchain = (host['Z'][:nrows], host['lp'][:nrows],
                                              host['zc'][:nrows])

    def stats_engine():
        """The engine, when the posterior statistics can be computed where the rows are:
        a run that started here (regular row layout) with every row in the local replica."""
        if resume or not hasattr(engine, 'posterior_summary'):
            return None
        return engine if (comm.size == 1 or comm.native) else None

    print("Yippee Ki Yay Monte Carlo!")
    log.msg(f"Start MCMC chains  ({time.ctime()})")
    log.msg("Lockstep cadence: " + ("strict (zlag = 0)" if not zlag_run else
            f"zlag = {zlag_run} (the other chains are seen {zlag_run} Z refreshes late)")
            + (f", chains sharded over {comm.size} ranks" if comm.size > 1 else ""))
    # ten progress reports: mcmc_driver.py:297-348
    nreport = 10
    edges = np.unique(np.ceil(np.arange(1, nreport + 1)*nzchain/nreport).astype(int))
    done = 0
    try:
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
                if overlap and done > 0:
                    try:
                        pull_rows(M0 + done*nchains, settled=True)    # (the segment before)
                    except RuntimeError:
                        overlap = False     # (rows not confirmed yet: the last pull takes them)
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
                ms.update_output(output, view, hsize, stats_engine())
                checkpoint.save(savefile, output, zlen)
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

    except BaseException:
        checkpoint.close()          # (no write left in flight, no half-written file handle)
        raise
    nrows = snapshot(final=True)
    pull_rows(nrows)
    checkpoint.wait()
    summary = ms.update_output(output, view, hsize, stats_engine())
    if _extras is not None and savefile is not None and comm.rank == 0:
        # (sample() adds its statistics and lets the same growing file take the final save)
        _extras['checkpoint'] = (checkpoint, zlen)
    else:
        checkpoint.close()
    if _extras is not None:
        _extras['summary'] = summary
        _extras['kde'] = getattr(engine, 'kde_pdf', None) if stats_engine() else None
    engine.close()

    Z = output['posterior']
    nsample = len(Z)*thinning
    nzsample = summary['n'] if summary is not None else 0
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
