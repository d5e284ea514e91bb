"""The maximum-a-posteriori pre-fit of ``sample(leastsq=...)``.

Runs once on the host before the chains start (mc3/fit_driver.py:15-206, called at
mc3/sampler_driver.py:413-440): it is not on the chain-update path, but ``sample()`` hands its
result to ``mcmc()`` as ``fit_output`` (the starting best fit, mcmc_driver.py:276-278) and
compares the chains' best fit with it afterwards, so it is kept as a small SciPy-backed host
function rather than refused.  Minimised: the vector of weighted data residuals followed by
the Gaussian-prior residuals (src_c/_chisq.c:37-79); 'lm' = Levenberg-Marquardt (MINPACK,
unbounded), 'trf' = trust-region reflective inside [pmin, pmax].
"""
import numpy as np
import scipy.optimize as so

__all__ = ['fit']

# (what the reference asks of MINPACK / trf: converge to rounding, fit_driver.py:171,183)
_TOL = 3e-16


def _weighted_residuals(full, model, data, uncert, prior, priorlow, priorup):
    """(model - data)/uncert, then (p - prior)/width for the parameters with a Gaussian
    prior (both widths positive), the width by the side of the prior p is on."""
    res = [(model - data)/uncert]
    gauss = np.flatnonzero((priorlow > 0) & (priorup > 0))
    if len(gauss):
        off = full[gauss] - prior[gauss]
        res.append(off/np.where(off > 0, priorup[gauss], priorlow[gauss]))
    return np.concatenate(res)


def fit(data, uncert, func, params, indparams=[], indparams_dict={},
        pstep=None, pmin=None, pmax=None, prior=None, priorlow=None, priorup=None,
        leastsq='lm'):
    """Best-fitting parameters of ``func`` to ``data``: the signature and the returned
    dictionary ('bestp', 'best_log_post', 'best_chisq', 'best_model', 'optimizer_res') of
    mc3.fit."""
    if leastsq not in ('lm', 'trf'):
        raise ValueError(f"Invalid 'leastsq' input ({leastsq}). Must select from "
                         "['lm', 'trf']")
    full = np.array(params, dtype=float)
    n = len(full)
    data = np.asarray(data, float)
    uncert = np.asarray(uncert, float)
    pstep = np.ones(n) if pstep is None else np.asarray(pstep, float)
    pmin = np.full(n, -np.inf) if pmin is None else np.asarray(pmin, float)
    pmax = np.full(n, np.inf) if pmax is None else np.asarray(pmax, float)
    if prior is None or priorlow is None or priorup is None:
        prior = priorlow = priorup = np.zeros(n)
    prior, priorlow, priorup = (np.asarray(v, float) for v in (prior, priorlow, priorup))
    ifree = np.flatnonzero(pstep > 0)
    shares = [(j, -int(pstep[j]) - 1) for j in np.flatnonzero(pstep < 0)]

    def expand(free):
        full[ifree] = free
        for j, src in shares:                     # mc3/chain.py:246-247
            full[j] = full[src]
        return full

    def objective(free):
        p = expand(free)
        model = func(p, *indparams, **indparams_dict)
        return _weighted_residuals(p, model, data, uncert, prior, priorlow, priorup)

    if leastsq == 'lm':
        res = so.least_squares(objective, full[ifree], method='lm',
                               ftol=_TOL, xtol=_TOL, gtol=_TOL)
    else:
        res = so.least_squares(objective, full[ifree], method='trf',
                               bounds=(pmin[ifree], pmax[ifree]),
                               ftol=_TOL, xtol=_TOL, gtol=_TOL)
    bestp = np.copy(expand(res.x))
    resid = objective(res.x)
    best_log_post = -0.5*float(np.sum(resid**2))
    from . import stats as ms
    log_prior = ms.log_prior(bestp[ifree], prior, priorlow, priorup, pstep)
    return {
        'bestp': bestp,
        'best_log_post': best_log_post,
        'best_chisq': -2.0*(best_log_post - float(log_prior)),
        'best_model': func(bestp, *indparams, **indparams_dict),
        'optimizer_res': res,
    }
