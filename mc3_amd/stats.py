"""Host-side statistics that assemble mc3's output dictionary.

Mirrors what ``mcmc()``/``sample()`` call after the chains have run
(mc3/stats/stats.py: log_prior :287-392, update_output :805-852,
calc_bestfit_statistics :855-873, calc_sample_statistics :876-964,
marginal_statistics :667-802, cred_region :395-467; mc3/stats/gelman.py).  These
are post-run.  With the chains on a device the passes over the whole posterior --
mean, standard deviation, median and central percentiles, chisq = -2 (log_post -
log_prior), the kernel density behind the HPD statistics -- run where the rows are
(``summarise`` / ``update_output`` take the engine); the NumPy
forms below serve the CPU checker and arbitrary arrays.
"""
import numpy as np

from . import utils as mu


def log_prior(posterior, prior, priorlow, priorup, pstep):
    """-0.5 * sum of squared prior residuals per sample (stats.py:287-392):
    Gaussian priors where priorlow>0 and priorup>0 (two-sided widths), the
    reference's log-uniform branch where priorlow<0, nothing otherwise."""
    posterior = np.atleast_2d(np.asarray(posterior, float))
    prior, priorlow, priorup, pstep = (np.asarray(a, float) for a in
                                       (prior, priorlow, priorup, pstep))
    ifree = np.flatnonzero(pstep > 0)
    delta = posterior - prior[ifree]
    for i, j in enumerate(ifree):
        if priorlow[j] > 0 and priorup[j] > 0:
            col = delta[:, i]
            width = np.where(col < 0, priorlow[j], priorup[j])
            delta[:, i] = np.where(col == 0, 0.0, col/width)
        elif priorlow[j] < 0:
            delta[:, i] = 2.0*np.log(posterior[:, i])
        else:
            delta[:, i] = 0.0
    logp = -0.5*np.sum(delta**2, axis=1)
    return logp[0] if np.size(logp) == 1 else logp


# ---- the L1 likelihood functions (mc3.stats.chisq / residuals / dwt_chisq /
# dwt_daub4) on arbitrary model arrays, evaluated on the device --------------------
def _prior_args(params, priors, priorlow, priorup):
    """The Gaussian-prior selection of stats.py:211-216."""
    if params is None or priors is None or priorlow is None or priorup is None:
        return None, None, None, 0
    params, priors, priorlow, priorup = (np.asarray(a, float) for a in
                                         (params, priors, priorlow, priorup))
    ip = (priorlow > 0) & (priorup > 0)
    return (np.ascontiguousarray((params - priors)[ip]), np.ascontiguousarray(priorlow[ip]),
            np.ascontiguousarray(priorup[ip]), int(np.sum(ip)))


def residuals(model, data, uncert, params=None, priors=None, priorlow=None,
              priorup=None, device=0):
    """mc3.stats.residuals (stats.py:97-147 -> src_c/_chisq.c:37-79)."""
    from . import _lib
    m, d, u = (_lib.f64(a) for a in (model, data, uncert))
    off, lo, up, npri = _prior_args(params, priors, priorlow, priorup)
    out = np.empty(len(m) + npri)
    _lib.check(_lib.load().mc3hip_residuals(
        device, _lib.dptr(m), _lib.dptr(d), _lib.dptr(u), len(m), _lib.dptr(off),
        _lib.dptr(lo), _lib.dptr(up), npri, _lib.dptr(out)))
    return out


def chisq(model, data, uncert, params=None, priors=None, priorlow=None,
          priorup=None, device=0):
    """mc3.stats.chisq (stats.py:150-216 -> src_c/_chisq.c:111-140)."""
    import ctypes
    from . import _lib
    m, d, u = (_lib.f64(a) for a in (model, data, uncert))
    off, lo, up, npri = _prior_args(params, priors, priorlow, priorup)
    out = ctypes.c_double()
    _lib.check(_lib.load().mc3hip_chisq(
        device, _lib.dptr(m), _lib.dptr(d), _lib.dptr(u), len(m), _lib.dptr(off),
        _lib.dptr(lo), _lib.dptr(up), npri,
        ctypes.cast(ctypes.byref(out), ctypes.POINTER(ctypes.c_double))))
    return out.value


def dwt_chisq(model, data, params, priors=None, priorlow=None, priorup=None, device=0):
    """mc3.stats.dwt_chisq (stats.py:219-284 -> src_c/_dwt.c:56-119)."""
    import ctypes
    from . import _lib
    if len(params) < 3:
        raise ValueError('Wavelet chisq should have at least three parameters')
    m, d, p = (_lib.f64(a) for a in (model, data, params))
    off, lo, up, npri = _prior_args(p, priors, priorlow, priorup)
    out = ctypes.c_double()
    _lib.check(_lib.load().mc3hip_dwt_chisq(
        device, _lib.dptr(p), len(p), _lib.dptr(m), _lib.dptr(d), len(d), _lib.dptr(off),
        _lib.dptr(lo), _lib.dptr(up), npri,
        ctypes.cast(ctypes.byref(out), ctypes.POINTER(ctypes.c_double))))
    return out.value


def dwt_daub4(array, inverse=False, device=0):
    """mc3.stats.dwt_daub4 (src_c/_dwt.c:154-186)."""
    from . import engine
    return engine.dwt_daub4(array, inverse, device)


def kde_grid(sample, mean, std, amin, amax, kde=None):
    """The density grid of cred_region (stats.py:440-452): a Gaussian KDE (Scott's
    bandwidth, as scipy.stats.gaussian_kde) of ``sample`` evaluated at 100 points
    between max(mean - 6 std, min) and min(mean + 6 std, max), interpolated linearly to
    3000.  ``kde(sample, x, cov)`` evaluates the density (device); default: NumPy."""
    import scipy.interpolate as si
    lo = max(mean - 6*std, amin)
    hi = min(mean + 6*std, amax)
    x = np.linspace(lo, hi, 100)
    xpdf = np.linspace(lo, hi, 3000)
    n = len(sample)
    cov = np.var(sample, ddof=1)*n**(-0.4)          # Scott: factor n^(-1/5), squared
    if kde is not None:
        dens = kde(sample, x, cov)
    else:
        d = x[:, None] - sample[None, :]
        dens = np.sum(np.exp(-0.5*d*d/cov), axis=1)/(n*np.sqrt(2*np.pi*cov))
    return si.interp1d(x, dens)(xpdf), xpdf


def hpd_level(pdf, quantile):
    """Density level of the highest-posterior-density region holding ``quantile`` of the
    probability (stats.py:454-467)."""
    order = np.argsort(pdf)[::-1]
    cdf = np.cumsum(pdf[order])
    ihpd = np.flatnonzero(cdf >= quantile*cdf[-1])[0]
    return np.amin(pdf[order][0:ihpd])


def cred_region(posterior=None, quantile=0.6827, pdf=None, xpdf=None):
    """Highest-posterior-density region from a Gaussian-KDE of a 1D sample
    (stats.py:395-467): returns (pdf, xpdf, HPDmin)."""
    if pdf is None and xpdf is None:
        thin = max(1, int(np.size(posterior)/120000))
        pdf, xpdf = kde_grid(posterior[::thin], np.mean(posterior), np.std(posterior),
                             np.amin(posterior), np.amax(posterior))
    if quantile is None:
        return pdf, xpdf, 0.0
    return pdf, xpdf, hpd_level(pdf, quantile)


def percentiles_from_order(n, percents, fetch):
    """np.percentile(..., method='linear') from exact order statistics: ``fetch(ranks)``
    returns the values at the 0-based ranks ([nranks, ...]).  NumPy's own arithmetic
    (virtual index, neighbours, _lerp), so that equal order statistics give equal
    percentiles bit for bit."""
    q = np.true_divide(np.asarray(percents, float), 100)
    virtual = (n - 1)*q                       # numpy's 'linear' method
    prev = np.floor(virtual).astype(np.int64)
    nxt = np.minimum(prev + 1, n - 1)
    prev = np.clip(prev, 0, n - 1)
    ranks = np.unique(np.concatenate([prev, nxt]))
    vals = fetch(ranks)
    out = []
    for vi, lo, hi in zip(virtual, prev, nxt):
        a = vals[np.searchsorted(ranks, lo)]
        b = vals[np.searchsorted(ranks, hi)]
        t = vi - np.floor(vi)
        diff = b - a
        out.append(np.where(t >= 0.5, b - diff*(1 - t), a + diff*t))
    return np.array(out)


def summarise(posterior=None, engine=None, burn=0, quantile=0.683, nrefresh=None):
    """n, mean, std, min, max, median and the central ``quantile`` bounds of every free
    parameter of the burned posterior: from the device when ``engine`` offers it (the rows
    of refreshes [burn, done), wherever they are), else from the array."""
    pct = [50*(1 - quantile), 50.0, 50*(1 + quantile)]
    if engine is not None and hasattr(engine, 'posterior_summary'):
        n = ((nrefresh or engine.refresh_done) - burn)*engine.nchains
        got = {}

        def fetch(ranks):
            got.update(engine.posterior_summary(burn, ranks, nrefresh))
            return got['order']
        low, med, high = percentiles_from_order(n, pct, fetch)
        return {'n': n, 'mean': got['mean'], 'std': np.sqrt(got['var']), 'min': got['min'],
                'max': got['max'], 'median': med, 'low': low, 'high': high}
    low, med, high = np.percentile(posterior, pct, axis=0)
    return {'n': len(posterior), 'mean': np.mean(posterior, axis=0),
            'std': np.std(posterior, axis=0), 'min': np.amin(posterior, axis=0),
            'max': np.amax(posterior, axis=0), 'median': med, 'low': low, 'high': high}


def marginal_statistics(posterior, statistics='med_central', quantile=0.683,
                        pdf=None, xpdf=None):
    """Per-parameter estimate and credible interval (stats.py:667-802)."""
    nsamples, nparams = np.shape(posterior)
    values = np.full(nparams, np.nan)
    low = np.full(nparams, np.nan)
    high = np.full(nparams, np.nan)
    if statistics is None:
        return values, low, high
    if pdf is None or xpdf is None:
        pdf = [None]*nparams
        xpdf = [None]*nparams
    if statistics == 'med_central' and quantile is not None:
        # median and both bounds from a single selection pass
        low, values, high = np.percentile(
            posterior, [50*(1 - quantile), 50.0, 50*(1 + quantile)], axis=0)
        return values, low, high
    if statistics.startswith('med_'):
        values = np.median(posterior, axis=0)
    elif statistics.startswith('max_'):
        for i in range(nparams):
            pdf[i], xpdf[i], _ = cred_region(posterior[:, i], quantile, pdf[i], xpdf[i])
            values[i] = xpdf[i][np.argmax(pdf[i])]
    if quantile is None:
        return values, low, high
    if statistics.endswith('_central'):
        low = np.percentile(posterior, 50*(1 - quantile), axis=0)
        high = np.percentile(posterior, 50*(1 + quantile), axis=0)
    elif statistics.endswith('_like'):
        for i in range(nparams):
            pdf[i], xpdf[i], hmin = cred_region(posterior[:, i], quantile, pdf[i], xpdf[i])
            inside = xpdf[i][pdf[i] > hmin]
            low[i], high[i] = np.amin(inside), np.amax(inside)
    return values, low, high


def calc_sample_statistics(posterior, bestp, pstep, quantile=0.683,
                           calc_hpd=False, pdf=None, xpdf=None):
    """median, mean, std, central bounds [, mode, HPD bounds] over all parameters;
    fixed parameters keep bestp, shared ones copy their source (stats.py:876-964)."""
    pstep = np.asarray(pstep, float)
    ifree = np.flatnonzero(pstep > 0)
    ishare = np.flatnonzero(pstep < 0)
    out = {name: np.copy(bestp) for name in
           ('median', 'mean', 'low', 'high', 'mode', 'hlow', 'hhigh')}
    std = np.zeros(len(pstep))
    med, lo, hi = marginal_statistics(posterior, 'med_central', quantile)
    out['median'][ifree], out['low'][ifree], out['high'][ifree] = med, lo, hi
    out['mean'][ifree] = np.mean(posterior, axis=0)
    std[ifree] = np.std(posterior, axis=0)
    if calc_hpd:
        mode, hlo, hhi = marginal_statistics(posterior, 'max_like', quantile, pdf, xpdf)
        out['mode'][ifree], out['hlow'][ifree], out['hhigh'][ifree] = mode, hlo, hhi
    for i in ishare:
        j = -int(pstep[i]) - 1
        for arr in list(out.values()) + [std]:
            arr[i] = arr[j]
    base = (out['median'], out['mean'], std, out['low'], out['high'])
    if not calc_hpd:
        return base
    return base + (out['mode'], out['hlow'], out['hhigh'])


def spread_to_all(free_values, bestp, pstep):
    """[nfree] statistics -> [nparams]: fixed parameters keep bestp, shared ones copy
    their source (stats.py:938-949)."""
    pstep = np.asarray(pstep, float)
    out = np.copy(bestp)
    out[pstep > 0] = free_values
    for i in np.flatnonzero(pstep < 0):
        out[i] = out[-int(pstep[i]) - 1]
    return out


def statistics_from_summary(summary, bestp, pstep):
    """(medianp, meanp, stdp, median_low_bounds, median_high_bounds) of
    calc_sample_statistics (stats.py:876-964) from a ``summarise`` result."""
    std = spread_to_all(summary['std'], np.zeros(len(bestp)), pstep)
    return (spread_to_all(summary['median'], bestp, pstep),
            spread_to_all(summary['mean'], bestp, pstep), std,
            spread_to_all(summary['low'], bestp, pstep),
            spread_to_all(summary['high'], bestp, pstep))


def hpd_statistics(Z, zmask, summary, bestp, pstep, quantile=0.683, kde=None):
    """The KDE half of calc_sample_statistics(calc_hpd=True) (stats.py:909-935): marginal
    mode and highest-posterior-density bounds over all parameters.  Only the rows the
    density is built from -- every thin-th row of the burned posterior Z[zmask], as
    cred_region thins it -- are touched; the moments come from ``summary``."""
    thin = max(1, int(len(zmask)/120000))
    rows = Z[zmask[::thin]]
    nfree = rows.shape[1]
    mode, lo, hi = np.empty(nfree), np.empty(nfree), np.empty(nfree)
    for i in range(nfree):
        pdf, xpdf = kde_grid(np.ascontiguousarray(rows[:, i]), summary['mean'][i],
                             summary['std'][i], summary['min'][i], summary['max'][i], kde)
        mode[i] = xpdf[np.argmax(pdf)]
        inside = xpdf[pdf > hpd_level(pdf, quantile)]
        lo[i], hi[i] = np.amin(inside), np.amax(inside)
    return (spread_to_all(mode, bestp, pstep), spread_to_all(lo, bestp, pstep),
            spread_to_all(hi, bestp, pstep))


def calc_bestfit_statistics(bestp, chain):
    """best chisq, reduced chisq, BIC, log-posterior, model and residual scatter
    at the best-fitting parameters (stats.py:855-873)."""
    ndata = len(chain.data)
    best_model, opt_chisq = chain.eval_model(bestp, ret='both')
    best_log_post = -0.5*opt_chisq
    best_log_prior = log_prior(bestp[chain.ifree], chain.prior, chain.priorlow,
                               chain.priorup, chain.pstep)
    best_chisq = -2*(best_log_post - best_log_prior)
    bic = best_chisq + chain.nfree*np.log(ndata)
    red_chisq = np.nan if ndata <= chain.nfree else best_chisq/(ndata - chain.nfree)
    return (best_chisq, red_chisq, bic, best_log_post, best_model,
            np.std(best_model - chain.data))


def update_output(output, chain, hsize, engine=None):
    """Fill the output dictionary from a Chain-shaped object (stats.py:805-852).
    ``engine``: the device engine the rows came from -- chisq and the statistics of the
    burned posterior are then computed on the device instead of over the host copies.
    Returns the ``summarise`` result of the burned posterior (None before every chain has
    passed its burn-in)."""
    zburn = output['burnin']
    zchain = chain.zchain
    first = int(np.argmax(zchain >= 0)) if np.any(zchain >= 0) else len(zchain)
    if np.all(zchain[:first] < 0) and np.all(zchain[first:] >= 0):
        valid = slice(first, len(zchain))        # (the usual layout: views, no copies)
        nvalid = len(zchain) - first
    else:
        valid = zchain >= 0
        nvalid = int(np.sum(valid))
    nsample = nvalid*chain.thinning
    Z = chain.Z[valid]
    lp = chain.log_post[valid]
    output['posterior'] = Z
    output['zchain'] = zchain[valid]
    on_device = engine is not None and hasattr(engine, 'chisq_rows') and \
        isinstance(valid, slice) and first == engine.M0
    if on_device:
        output['chisq'] = engine.chisq_rows(first, nvalid)
    else:
        output['chisq'] = -2.0*(lp - log_prior(Z, chain.prior, chain.priorlow,
                                               chain.priorup, chain.pstep))
    output['log_post'] = lp
    output['acceptance_rate'] = chain.numaccept.value*100.0/nsample
    (output['best_chisq'], output['red_chisq'], output['BIC'],
     output['best_log_post'], output['best_model'],
     output['stddev_residuals']) = calc_bestfit_statistics(chain.bestp, chain)
    output['bestp'] = chain.bestp
    if not np.all(chain.chainsize > (zburn + hsize)):
        return None
    # (rows that came from the device engine of a run that started here are in the regular
    # layout by construction: the scan over zchain that would establish it is skipped)
    zmask = mu.burn_mask(output['zchain'], zburn,
                         layout=(0, engine.nchains) if on_device else None)
    regular = on_device and len(zmask) == (engine.refresh_done - zburn)*engine.nchains
    if regular:
        summary = summarise(engine=engine, burn=zburn)
    else:
        summary = summarise(Z[zmask])
    output['zmask'] = zmask
    (output['medianp'], output['meanp'], output['stdp'],
     output['median_low_bounds'], output['median_high_bounds']) = \
        statistics_from_summary(summary, chain.bestp, chain.pstep)
    return summary


def gelman_rubin_from_moments(mean, var, niter):
    """PSRF from per-chain means and (population) variances over ``niter`` samples
    each -- psrf() of mc3/stats/gelman.py:64-92 with the reductions done elsewhere
    (mc3_amd.engine.DeviceChains.chain_moments)."""
    nchains = len(mean)
    W = np.mean(var, axis=0)
    B = niter/(nchains - 1.0)*np.sum((mean - np.mean(mean, axis=0))**2, axis=0)
    V = W*(niter - 1.0)/niter + B*(nchains + 1.0)/(niter*nchains)
    return np.sqrt(V/W)


def gelman_rubin(Z, zchain, burnin):
    """Potential scale reduction factor per parameter (mc3/stats/gelman.py:12-92)."""
    Z = np.asarray(Z)
    zchain = np.asarray(zchain)
    nchains = int(np.amax(zchain)) + 1
    npars = Z.shape[1]
    # Fast path for the regular layout (rows in repeating chain order after the
    # initial population): one reshape instead of a scan per chain.
    first = int(np.argmax(zchain >= 0))
    body = zchain[first:]
    if len(body) % nchains == 0 and np.all(zchain[:first] < 0) and \
            np.array_equal(body[:nchains], np.arange(nchains)) and \
            np.array_equal(body[-nchains:], np.arange(nchains)) and \
            np.array_equal(body.reshape(-1, nchains)[:, 0], np.zeros(len(body)//nchains, body.dtype)):
        niter = len(body)//nchains - burnin
        if niter < 1:
            print('Not enough samples for Gelman-Rubin test.')
            return np.zeros(npars)
        chains = Z[first:].reshape(-1, nchains, npars)[burnin:]     # [iter, chain, par]
        W = np.mean(np.var(chains, axis=0), axis=0)
        means = np.mean(chains, axis=0)
        B = niter/(nchains - 1.0)*np.sum((means - np.mean(means, axis=0))**2, axis=0)
        V = W*(niter - 1.0)/niter + B*(nchains + 1.0)/(niter*nchains)
        return np.sqrt(V/W)
    counts = np.array([np.sum(zchain == c) for c in range(nchains)]) - burnin
    niter = int(np.amin(counts))
    if niter < 1:
        print('Not enough samples for Gelman-Rubin test.')
        return np.zeros(npars)
    chains = np.empty((nchains, niter, npars))
    for c in range(nchains):
        chains[c] = Z[np.flatnonzero(zchain == c)[burnin:burnin + niter]]
    W = np.mean(np.var(chains, axis=1), axis=0)
    means = np.mean(chains, axis=1)
    B = niter/(nchains - 1.0)*np.sum((means - np.mean(means, axis=0))**2, axis=0)
    V = W*(niter - 1.0)/niter + B*(nchains + 1.0)/(niter*nchains)
    return np.sqrt(V/W)
