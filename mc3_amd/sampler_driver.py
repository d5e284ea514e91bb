"""sample(): mc3's user-facing driver with the chains on an MI355X.

Same signature, defaults, argument checks, error messages and output
dictionary as ``mc3.sample`` (mc3/sampler_driver.py:25-601); the call to
``mcmc()`` at :448-457 goes to mc3_amd.mcmc_driver.mcmc.  What is outside the
hot path is NOT re-implemented: the least-squares pre-fit (``leastsq``), plots,
themes and the text summary file are delegated to an installed ``mc3`` when one
is importable, and refused / skipped with a clear message otherwise
(SURVEY.md section 2).
"""
import importlib
import os
import sys

import numpy as np

from . import stats as ms
from . import utils as mu
from .mcmc_driver import mcmc

__all__ = ['sample']


def _reference():
    try:
        import mc3
        return mc3
    except Exception:
        return None


def sample(
        data=None, uncert=None, func=None, params=None, indparams=[],
        indparams_dict={},
        pmin=None, pmax=None, pstep=None,
        prior=None, priorlow=None, priorup=None,
        sampler=None, ncpu=None, leastsq=None, chisqscale=False,
        nchains=7, nsamples=None, burnin=0, thinning=1,
        grtest=True, grbreak=0.0, grnmin=0.5, wlike=False,
        fgamma=1.0, fepsilon=0.0, hsize=10, kickoff='normal',
        plots=False, theme='blue', statistics='med_central',
        ioff=False, showbp=True, savefile=None, resume=False,
        rms=False, log=None, pnames=None, texnames=None,
        **kwargs):
    """Model-fitting driver (see mc3.sample for the argument documentation).

    Differences from the reference:
    - ``func`` must belong to the device model family: mc3_amd.models.Polynomial /
      Linear, or any callable that is affine in its parameters (recognised by
      probing); anything else raises ValueError.
    - ``ncpu`` is ignored (the GPU advances all chains in lockstep).
    - extra keywords: ``seed`` (int), ``device`` (int), ``comm``
      (mc3_amd.parallel communicator for multi-GPU runs).
    """
    from . import __version__
    if isinstance(log, str):
        log = mu.Log(log, append=resume)
        closelog = True
    else:
        closelog = False
        if log is None:
            log = mu.Log()

    log.msg(
        f"\n{log.sep}\n"
        "  mc3 chain-update path on AMD MI355X (mc3_amd).\n"
        f"  Version {__version__}.\n"
        f"{log.sep}\n\n")

    if sampler is None:
        log.error("'sampler' is a required argument")
    if nsamples is None and sampler in ['MRW', 'DEMC', 'snooker']:
        log.error("'nsamples' is a required argument for MCMC runs")
    if leastsq not in [None, 'lm', 'trf']:
        log.error(
            f"Invalid 'leastsq' input ({leastsq}). Must select from "
            "['lm', 'trf']")
    if sampler not in ['mrw', 'demc', 'snooker']:
        log.error(f"Invalid 'sampler' input ({sampler}). Must select from "
                  "['mrw', 'demc', 'snooker']")
    if nsamples is None:
        log.error("'nsamples' is a required argument for MCMC runs")

    if params is None:
        log.error("'params' is a required argument")
    if isinstance(params, str):
        log.error("reading 'params' from a file is outside the MI355X path; "
                  "pass an array")
    # unpack a 2D params input: sampler_driver.py:284-299
    if np.ndim(params) > 1:
        ninfo, _ = np.shape(params)
        if ninfo == 7:
            prior, priorlow, priorup = params[4], params[5], params[6]
        if ninfo >= 4:
            pstep = params[3]
        if ninfo >= 3:
            pmin, pmax = params[1], params[2]
        else:
            log.error('Invalid format/shape for params input file')
        params = params[0]
    params = np.array(params, float)

    if data is None:
        log.error("'data' is a required argument")
    if isinstance(data, str):
        log.error("reading 'data' from a file is outside the MI355X path; "
                  "pass an array")
    if np.ndim(data) > 1:
        data, uncert = data
    if uncert is None:
        log.error("'uncert' is a required argument")
    data = np.asarray(data, float)
    uncert = np.array(uncert, float)      # a copy: chisqscale rescales it

    resume = resume and (savefile is not None)
    if resume:
        log.msg(f"\n\n{log.sep}\n{log.sep}  Resuming previous MCMC run.\n\n")

    # the model function: sampler_driver.py:321-331
    if isinstance(func, (list, tuple, np.ndarray)):
        if len(func) == 3:
            sys.path.append(func[2])
        else:
            sys.path.append(os.getcwd())
        func = getattr(importlib.import_module(func[1]), func[0])
    elif not callable(func):
        log.error(
            "'func' must be either a callable or an iterable of strings "
            "with the model function, file, and path names")

    nparams = len(params)
    ndata = len(data)
    if pnames is None and texnames is not None:
        pnames = texnames
    elif pnames is not None and texnames is None:
        texnames = pnames
    elif pnames is None and texnames is None:
        pnames = texnames = mu.default_parnames(nparams)
    pnames = np.asarray(pnames)
    texnames = np.asarray(texnames)

    pmin = np.tile(-np.inf, nparams) if pmin is None else np.asarray(pmin, float)
    pmax = np.tile(np.inf, nparams) if pmax is None else np.asarray(pmax, float)
    pstep = 0.1*np.abs(params) if pstep is None else np.asarray(pstep, float)
    if prior is None or priorup is None or priorlow is None:
        prior = np.zeros(nparams)
        priorlow = np.zeros(nparams)
        priorup = np.zeros(nparams)
    prior = np.array(prior, float)
    priorlow = np.array(priorlow, float)
    priorup = np.array(priorup, float)
    # priors only act on free parameters: sampler_driver.py:371-373
    priorlow[pstep <= 0] = 0.0
    priorup[pstep <= 0] = 0.0

    if np.any(params < pmin) or np.any(params > pmax):
        pout = ""
        for pname, par, minp, maxp in zip(pnames, params, pmin, pmax):
            if par < minp:
                pout += f"\n{pname[:11]:11s}  {minp: 12.5e} < {par: 12.5e}"
            if par > maxp:
                pout += f"\n{pname[:11]:26s}  {par: 12.5e} > {maxp: 12.5e}"
        log.error(
            "Some initial-guess values are out of bounds:\n"
            "Param name           pmin          value           pmax\n"
            "-----------  ------------   ------------   ------------"
            f"{pout}")

    nfree = int(np.sum(pstep > 0))
    ifree = np.where(pstep > 0)[0]
    ishare = np.where(pstep < 0)[0]

    model0 = func(params, *indparams, **indparams_dict)
    if np.shape(model0) != np.shape(data):
        log.error(
            f"The size of the data array ({np.size(data)}) does not "
            f"match the size of the func() output ({np.size(model0)})")

    if savefile is not None:
        fpath, _ = os.path.split(os.path.realpath(savefile))
        if not os.path.exists(fpath):
            log.warning(
                f"Output folder path: '{fpath}' does not exist. "
                "Creating new folder.")
            os.makedirs(fpath)

    ref = _reference()
    chisq_factor = 1.0
    fit_output = None
    if leastsq is not None:
        if ref is None:
            log.error(
                "leastsq pre-fitting (mc3.fit, SciPy-bound) is outside the "
                "MI355X path and no 'mc3' installation is importable to "
                "delegate it to; run with leastsq=None", NotImplementedError)
        fit_output = ref.fit(
            data, uncert, func, np.copy(params), indparams, indparams_dict,
            pstep, pmin, pmax, prior, priorlow, priorup, leastsq)
        log.msg("Least-squares best-fitting parameters:\n"
                f"  {fit_output['bestp']}\n\n", si=2)
        if chisqscale:
            chisq_factor = np.sqrt(fit_output['best_chisq']/(ndata - nfree))
            uncert *= chisq_factor
            fit_output = ref.fit(
                data, uncert, func, np.copy(params), indparams, indparams_dict,
                pstep, pmin, pmax, prior, priorlow, priorup, leastsq)
            log.msg("Least-squares best-fitting parameters (rescaled chisq):"
                    f"\n  {fit_output['bestp']}\n\n", si=2)
        params = np.copy(fit_output['bestp'])

    if resume:
        with np.load(savefile) as oldrun:
            uncert *= float(oldrun['chisq_factor'])/chisq_factor
            chisq_factor = float(oldrun['chisq_factor'])

    extra = {k: kwargs[k] for k in ('seed', 'device', 'comm', 'engine_factory')
             if k in kwargs}
    output = mcmc(
        data, uncert, func, params, indparams, indparams_dict,
        pmin, pmax, pstep, prior, priorlow, priorup, nchains, ncpu, nsamples,
        sampler, wlike, fit_output, grtest, grbreak, grnmin, burnin, thinning,
        fgamma, fepsilon, hsize, kickoff, savefile, resume, log,
        pnames, texnames, **extra)

    output['chisq_factor'] = chisq_factor

    if leastsq is not None:
        delta_log_post = output['best_log_post'] - fit_output['best_log_post']
        delta_pars = output['bestp'] - fit_output['bestp']
        if delta_log_post > 5.0e-8 and np.any(delta_pars != 0.0):
            log.warning(
                "MCMC found a better fit than the minimizer:\n"
                "MCMC best-fitting parameters:        (chisq={:.8g})\n{}\n"
                "Minimizer best-fitting parameters:   (chisq={:.8g})\n{}".
                format(-2*output['best_log_post'], output['bestp'],
                       -2*fit_output['best_log_post'], fit_output['bestp']))

    # sample statistics without the burn-in: sampler_driver.py:475-501
    posterior, zchain, zmask = mu.burn(
        Z=output['posterior'], zchain=output['zchain'], burnin=output['burnin'])
    bestp = output['bestp']
    # the median / central-interval statistics of this same burned posterior were
    # computed by update_output (sampler_driver.py:490-497 recomputes them); only the
    # KDE-based mode and HPD bounds are added here
    if 'medianp' in output and 'zmask' in output and len(output['zmask']) == len(zmask):
        mode, hlo, hhi = ms.hpd_statistics(posterior, bestp, pstep)
    else:
        stats = ms.calc_sample_statistics(posterior, bestp, pstep, calc_hpd=True)
        (output['medianp'], output['meanp'], output['stdp'],
         output['median_low_bounds'], output['median_high_bounds'],
         mode, hlo, hhi) = stats
    output['mode'], output['hpd_low_bounds'], output['hpd_high_bounds'] = mode, hlo, hhi
    output['CRlo'] = output['hpd_low_bounds'] - bestp
    output['CRhi'] = output['hpd_high_bounds'] - bestp
    output['CRlo'][pstep == 0] = output['CRhi'][pstep == 0] = 0.0

    median, stdp = output['medianp'], output['stdp']
    log.msg(
        "\nParameter name     best fit   median      1sigma_low   1sigma_hi        S/N"
        "\n--------------- -----------  -----------------------------------  ---------",
        width=80)
    for i in range(nparams):
        lo = output['median_low_bounds'][i] - median[i]
        hi = output['median_high_bounds'][i] - median[i]
        if i in ifree:
            snr = f"{np.abs(bestp[i])/stdp[i]:.1f}"
        elif i in ishare:
            snr = f"[share{-int(pstep[i]):02d}]"
        else:
            snr = "[fixed]"
            lo = hi = 0.0
        log.msg(
            f"{pnames[i][0:15]:<15} {bestp[i]:11.4e}  {median[i]:11.4e} "
            f"{lo:11.4e} {hi:11.4e}  {snr:>9s}", width=160)

    bic = output['BIC']
    fmt = len(f"{bic:.4f}")
    scale_txt = (f"sqrt(reduced chi-squared) factor: {chisq_factor:.4f}\n"
                 if chisqscale else '')
    log.msg(
        f"\n{scale_txt}"
        f"Best-parameter's chi-squared:       {output['best_chisq']:{fmt}.4f}\n"
        f"Best-parameter's -2*log(posterior): {-2.0*output['best_log_post']:{fmt}.4f}\n"
        f"Bayesian Information Criterion:     {bic:{fmt}.4f}\n"
        f"Reduced chi-squared:                {output['red_chisq']:{fmt}.4f}\n"
        f"Standard deviation of residuals:  {output['stddev_residuals']:.6g}\n",
        indent=2)

    if savefile is not None:
        np.savez(savefile, **output)
        log.msg("\nOutput sampler files:")
        log.msg(savefile, indent=2)

    if plots:
        if ref is None:
            log.warning("plots=True ignored: plotting is outside the MI355X "
                        "path and no 'mc3' installation is importable")
        else:
            root = os.path.splitext(savefile)[0] if savefile else 'mc3'
            ref.plots.trace(
                output['posterior'], zchain=output['zchain'],
                burnin=output['burnin'], pnames=texnames[ifree],
                savefile=f'{root}_trace.png')
            post = ref.plots.Posterior(
                posterior, pnames=texnames[ifree], theme=theme,
                bestp=bestp[ifree], statistics=statistics)
            post.plot(savefile=f'{root}_pairwise_posterior.png',
                      show_estimates=showbp)
            post.plot_histogram(savefile=f'{root}_marginal_posterior.png',
                                show_estimates=showbp)

    if closelog:
        log.msg(log.logname, indent=2)
        log.close()
    return output
