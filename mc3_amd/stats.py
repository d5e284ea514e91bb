"""Host-side statistics that assemble mc3's output dictionary.

Mirrors what ``mcmc()``/``sample()`` call after the chains have run
(mc3/stats/stats.py: log_prior :287-392, update_output :805-852,
calc_bestfit_statistics :855-873, calc_sample_statistics :876-964,
marginal_statistics :667-802, cred_region :395-467; mc3/stats/gelman.py).  These
are post-run, NumPy/SciPy-bound and outside the hot path; they are kept on the
host (SURVEY.md section 8f lists device versions as "next").
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


def cred_region(posterior=None, quantile=0.6827, pdf=None, xpdf=None):
    """Highest-posterior-density region from a Gaussian-KDE of a 1D sample
    (stats.py:395-467): returns (pdf, xpdf, HPDmin)."""
    import scipy.stats as ss
    import scipy.interpolate as si
    if pdf is None and xpdf is None:
        thin = max(1, int(np.size(posterior)/120000))
        kernel = ss.gaussian_kde(posterior[::thin])
        mean, std = np.mean(posterior), np.std(posterior)
        lo = max(mean - 6*std, np.amin(posterior))
        hi = min(mean + 6*std, np.amax(posterior))
        x = np.linspace(lo, hi, 100)
        xpdf = np.linspace(lo, hi, 3000)
        pdf = si.interp1d(x, kernel.evaluate(x))(xpdf)
    if quantile is None:
        return pdf, xpdf, 0.0
    order = np.argsort(pdf)[::-1]
    cdf = np.cumsum(pdf[order])
    ihpd = np.flatnonzero(cdf >= quantile*cdf[-1])[0]
    return pdf, xpdf, np.amin(pdf[order][0:ihpd])


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


def hpd_statistics(posterior, bestp, pstep, quantile=0.683):
    """The KDE half of calc_sample_statistics(calc_hpd=True): marginal mode and
    highest-posterior-density bounds over all parameters."""
    pstep = np.asarray(pstep, float)
    ifree = np.flatnonzero(pstep > 0)
    mode, lo, hi = np.copy(bestp), np.copy(bestp), np.copy(bestp)
    mode[ifree], lo[ifree], hi[ifree] = marginal_statistics(posterior, 'max_like', quantile)
    for i in np.flatnonzero(pstep < 0):
        j = -int(pstep[i]) - 1
        mode[i], lo[i], hi[i] = mode[j], lo[j], hi[j]
    return mode, lo, hi


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


def update_output(output, chain, hsize):
    """Fill the output dictionary from a Chain-shaped object (stats.py:805-852)."""
    zburn = output['burnin']
    valid = chain.zchain >= 0
    nsample = np.sum(valid)*chain.thinning
    Z = chain.Z[valid]
    lp = chain.log_post[valid]
    output['posterior'] = Z
    output['zchain'] = chain.zchain[valid]
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
    posterior, _, zmask = mu.burn(Z=Z, zchain=output['zchain'], burnin=zburn)
    stats = calc_sample_statistics(posterior, chain.bestp, chain.pstep)
    output['zmask'] = zmask
    (output['medianp'], output['meanp'], output['stdp'],
     output['median_low_bounds'], output['median_high_bounds']) = stats
    return posterior


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
