"""sample(): mc3's user-facing driver with the chains on an MI355X.

Same signature, defaults, error messages and output dictionary as ``mc3.sample``
(mc3/sampler_driver.py:25-601); the call to ``mcmc()`` at :448-457 goes to
mc3_amd.mcmc_driver.mcmc.  The least-squares pre-fit (``leastsq`` / ``chisqscale``,
mc3/fit_driver.py) runs once on the host through mc3_amd.fit, as the reference's does
through SciPy; what else lies outside the hot path is NOT re-implemented: plots are
skipped with a warning, parameter / data files are not read.

Layout: the argument handling is a pipeline of small normalisers over one
``_Inputs`` record, each stating the rule of the reference it implements; only the
error texts (which the reference's tests match on, tests/test_mcmc.py:328-409) are
shared with it.
"""
import importlib
import os
import sys

import functools

import numpy as np

from . import stats as ms
from . import utils as mu
from .fit_driver import fit
from .mcmc_driver import mcmc

__all__ = ['sample']

SAMPLERS = ['mrw', 'demc', 'snooker']


class _Inputs:
    """The array-valued arguments of sample() on their way to mcmc()."""
    __slots__ = ('data', 'uncert', 'params', 'pmin', 'pmax', 'pstep', 'prior',
                 'priorlow', 'priorup', 'pnames', 'texnames')


def _require(log, **named):
    """'x' is a required argument -- in the order given."""
    for name, value in named.items():
        if value is None:
            log.error(f"'{name}' is a required argument")


def _split_params(inp, log, params, pmin, pmax, pstep, prior, priorlow, priorup):
    """params may carry the other per-parameter arrays as extra rows, in the order
    (values, pmin, pmax, pstep, prior, priorlow, priorup): 3, 4 or 7 rows
    (sampler_driver.py:284-299).  Row-borne values win over keyword ones."""
    if isinstance(params, str):
        log.error("reading 'params' from a file is outside the MI355X path; "
                  "pass an array")
    table = np.atleast_2d(np.asarray(params, float))
    nrows = len(table) if np.ndim(params) > 1 else 1
    if nrows > 1:
        if nrows < 3:
            log.error('Invalid format/shape for params input file')
        pmin, pmax = table[1], table[2]
        if nrows >= 4:
            pstep = table[3]
        if nrows == 7:
            prior, priorlow, priorup = table[4], table[5], table[6]
    inp.params = np.array(table[0], float)
    n = len(inp.params)
    inp.pmin = np.full(n, -np.inf) if pmin is None else np.asarray(pmin, float)
    inp.pmax = np.full(n, np.inf) if pmax is None else np.asarray(pmax, float)
    inp.pstep = 0.1*np.abs(inp.params) if pstep is None else np.asarray(pstep, float)
    # priors: all three or none (sampler_driver.py:360-369), and only on free
    # parameters (:371-373)
    if prior is None or priorlow is None or priorup is None:
        prior = priorlow = priorup = np.zeros(n)
    inp.prior = np.array(prior, float)
    fixed_or_shared = inp.pstep <= 0
    inp.priorlow = np.where(fixed_or_shared, 0.0, np.asarray(priorlow, float))
    inp.priorup = np.where(fixed_or_shared, 0.0, np.asarray(priorup, float))


def _split_data(inp, log, data, uncert):
    """data may be a (data, uncert) pair (sampler_driver.py:301-312)."""
    if isinstance(data, str):
        log.error("reading 'data' from a file is outside the MI355X path; "
                  "pass an array")
    if np.ndim(data) > 1:
        data, uncert = data
    _require(log, uncert=uncert)
    inp.data = np.asarray(data, float)
    inp.uncert = np.array(uncert, float)


def _resolve_func(log, func):
    """A callable, or (name, module[, path]) strings (sampler_driver.py:321-331)."""
    if callable(func):
        return func
    if isinstance(func, (list, tuple, np.ndarray)):
        name, module = func[0], func[1]
        sys.path.append(func[2] if len(func) == 3 else os.getcwd())
        return getattr(importlib.import_module(module), name)
    log.error(
        "'func' must be either a callable or an iterable of strings "
        "with the model function, file, and path names")


def _names(inp, pnames, texnames):
    """Either name list stands in for the other; default names otherwise
    (sampler_driver.py:346-352)."""
    if pnames is None and texnames is None:
        pnames = mu.default_parnames(len(inp.params))
    inp.pnames = np.asarray(pnames if pnames is not None else texnames)
    inp.texnames = np.asarray(texnames if texnames is not None else pnames)


def _check_inside_bounds(inp, log):
    """The initial guess must lie inside [pmin, pmax] (sampler_driver.py:375-389)."""
    low = inp.params < inp.pmin
    high = inp.params > inp.pmax
    if not (np.any(low) or np.any(high)):
        return
    lines = []
    for j in np.flatnonzero(low | high):
        name = str(inp.pnames[j])[:11]
        if low[j]:
            lines.append(f"\n{name:11s}  {inp.pmin[j]: 12.5e} < {inp.params[j]: 12.5e}")
        if high[j]:
            lines.append(f"\n{name:26s}  {inp.params[j]: 12.5e} > {inp.pmax[j]: 12.5e}")
    log.error(
        "Some initial-guess values are out of bounds:\n"
        "Param name           pmin          value           pmax\n"
        "-----------  ------------   ------------   ------------"
        + "".join(lines))


def _summary_table(log, inp, output):
    """The closing parameter table and fit statistics (sampler_driver.py:503-545)."""
    bestp, median, stdp = output['bestp'], output['medianp'], output['stdp']
    lo = output['median_low_bounds'] - median
    hi = output['median_high_bounds'] - median
    log.msg(
        "\nParameter name     best fit   median      1sigma_low   1sigma_hi        S/N"
        "\n--------------- -----------  -----------------------------------  ---------",
        width=80)
    for j, step in enumerate(inp.pstep):
        if step > 0:
            tag = f"{np.abs(bestp[j])/stdp[j]:.1f}"
        elif step < 0:
            tag = f"[share{-int(step):02d}]"
        else:
            tag = "[fixed]"
            lo[j] = hi[j] = 0.0
        log.msg(
            f"{inp.pnames[j][0:15]:<15} {bestp[j]:11.4e}  {median[j]:11.4e} "
            f"{lo[j]:11.4e} {hi[j]:11.4e}  {tag:>9s}", width=160)
    bic = output['BIC']
    w = len(f"{bic:.4f}")
    log.msg(
        f"\nBest-parameter's chi-squared:       {output['best_chisq']:{w}.4f}\n"
        f"Best-parameter's -2*log(posterior): {-2.0*output['best_log_post']:{w}.4f}\n"
        f"Bayesian Information Criterion:     {bic:{w}.4f}\n"
        f"Reduced chi-squared:                {output['red_chisq']:{w}.4f}\n"
        f"Standard deviation of residuals:  {output['stddev_residuals']:.6g}\n",
        indent=2)


def host_threads():
    """Threads the host-side array math of a run may use: what this process may actually
    burn -- its CPU affinity AND its cgroup's CPU quota -- and never more than 8.

    Why the library cares (profiles/r06_e2e_stall.txt): the model callable is probed on the host
    and the statistics of the output are NumPy; on a box that SHOWS 256 CPUs but GRANTS 16
    (cgroup cpu.max 1600000 100000) OpenBLAS starts 64 threads for a matrix-vector product of a
    millisecond, the quota of the 100 ms period is gone at once and the kernel freezes every
    thread of the process for the rest of it -- the "~70 ms wait inside the HIP runtime" of rounds
    2-5, seen in whatever HIP call the main thread happened to be in.  sample() on C5 with 1e5
    samples: 100-200 ms per call before, 45 ms with the pools limited."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            with open(path) as f:
                words = f.read().split()
            if path.endswith('cpu.max'):
                if words and words[0] != 'max':
                    n = min(n, max(1, int(words[0])//int(words[1])))
            elif words and int(words[0]) > 0:
                with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                    n = min(n, max(1, int(words[0])//int(f.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(8, n))


def _sample(
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
    - ``leastsq`` / ``chisqscale``: the pre-fit runs on the host (mc3_amd.fit, SciPy, once,
      before the chains), with the reference's outputs; ``plots`` are outside the MI355X
      path and skipped with a warning.
    - extra keywords: ``seed`` (int), ``device`` (int), ``kickoff_stream`` ('philox':
      initial population drawn on the device, the default; 'numpy': from the global
      NumPy stream like the reference), ``comm``
      (mc3_amd.parallel communicator for multi-GPU runs: every rank makes the same
      call and gets the same dictionary; files are written by rank 0 only), ``zlag`` (the
      lockstep cadence: the other chains are seen that many refreshes late, include/mc3hip.h;
      default 0 = strict on any number of GPUs, 'auto' = parallel.CADENCE_ZLAG; recorded in
      ``output['zlag']``, the savefile and the log), ``wpad``
      (wavelet likelihood of a series whose length is no power of two: padded with zero
      residuals to the next one; without it such a series is refused -- mcmc()).
    """
    from . import __version__
    comm = kwargs.get('comm')
    lead = comm is None or comm.rank == 0        # the rank that owns the files
    own_log = isinstance(log, str)
    if own_log:
        log = mu.Log(log if lead else None, append=resume)
    elif log is None:
        log = mu.Log()
    log.msg(
        f"\n{log.sep}\n"
        "  mc3 chain-update path on AMD MI355X (mc3_amd).\n"
        f"  Version {__version__}.\n"
        f"{log.sep}\n\n")

    # -- what must be there, and the enumerations (sampler_driver.py:254-282) -------
    _require(log, sampler=sampler)
    if leastsq not in [None, 'lm', 'trf']:
        log.error(
            f"Invalid 'leastsq' input ({leastsq}). Must select from "
            "['lm', 'trf']")
    if sampler not in SAMPLERS:
        log.error(f"Invalid 'sampler' input ({sampler}). Must select from "
                  f"{SAMPLERS}")
    if nsamples is None:
        log.error("'nsamples' is a required argument for MCMC runs")
    _require(log, params=params)
    inp = _Inputs()
    _split_params(inp, log, params, pmin, pmax, pstep, prior, priorlow, priorup)
    _require(log, data=data)
    _split_data(inp, log, data, uncert)
    func = _resolve_func(log, func)
    _names(inp, pnames, texnames)
    _check_inside_bounds(inp, log)

    model0 = func(inp.params, *indparams, **indparams_dict)
    if np.shape(model0) != np.shape(inp.data):
        log.error(
            f"The size of the data array ({np.size(inp.data)}) does not "
            f"match the size of the func() output ({np.size(model0)})")

    resume = bool(resume) and savefile is not None
    # the least-squares pre-fit (sampler_driver.py:411-440): once, on the host; its best fit
    # is where the chains start.  chisqscale alone (leastsq=None) does nothing, as there
    chisq_factor = 1.0
    fit_output = None
    if leastsq is not None:
        def prefit():
            return fit(inp.data, inp.uncert, func, np.copy(inp.params), indparams,
                       indparams_dict, inp.pstep, inp.pmin, inp.pmax, inp.prior,
                       inp.priorlow, inp.priorup, leastsq)
        fit_output = prefit()
        log.msg(f"Least-squares best-fitting parameters:\n  {fit_output['bestp']}\n\n", si=2)
        if chisqscale:
            # data uncertainties scaled such that the reduced chisq of the best fit is 1
            nfree = int(np.sum(inp.pstep > 0))
            chisq_factor = np.sqrt(fit_output['best_chisq']/(len(inp.data) - nfree))
            inp.uncert *= chisq_factor
            fit_output = prefit()
            log.msg("Least-squares best-fitting parameters (rescaled chisq):"
                    f"\n  {fit_output['bestp']}\n\n", si=2)
        inp.params = np.copy(fit_output['bestp'])
    if resume:
        log.msg(f"\n\n{log.sep}\n{log.sep}  Resuming previous MCMC run.\n\n")
        with np.load(savefile) as previous:      # sampler_driver.py:442-445
            inp.uncert *= float(previous['chisq_factor'])/chisq_factor
            chisq_factor = float(previous['chisq_factor'])
    elif savefile is not None and lead:
        folder = os.path.dirname(os.path.realpath(savefile))
        if not os.path.exists(folder):
            log.warning(
                f"Output folder path: '{folder}' does not exist. "
                "Creating new folder.")
            os.makedirs(folder)

    extra = {k: kwargs[k] for k in ('seed', 'device', 'comm', 'engine_factory',
                                   'kickoff_stream', 'zlag', 'wpad') if k in kwargs}
    extra['_extras'] = side = {}
    output = mcmc(
        inp.data, inp.uncert, func, inp.params, indparams, indparams_dict,
        inp.pmin, inp.pmax, inp.pstep, inp.prior, inp.priorlow, inp.priorup,
        nchains, ncpu, nsamples, sampler, wlike, fit_output, grtest, grbreak, grnmin,
        burnin, thinning, fgamma, fepsilon, hsize, kickoff, savefile, resume, log,
        inp.pnames, inp.texnames, **extra)
    output['chisq_factor'] = chisq_factor
    if fit_output is not None:                   # sampler_driver.py:462-473
        better = output['best_log_post'] - fit_output['best_log_post']
        if better > 5.0e-8 and np.any(output['bestp'] != fit_output['bestp']):
            log.warning(
                "MCMC found a better fit than the minimizer:\n"
                f"MCMC best-fitting parameters:        (chisq={-2*output['best_log_post']:.8g})\n"
                f"{output['bestp']}\n"
                f"Minimizer best-fitting parameters:   "
                f"(chisq={-2*fit_output['best_log_post']:.8g})\n{fit_output['bestp']}")

    # -- statistics of the burned sample (sampler_driver.py:475-501): the median and
    # central-interval numbers were computed with the output dictionary (update_output,
    # on the device); the KDE mode and HPD bounds are added here from every thin-th burned
    # row and the moments already known -- the burned posterior is never materialised
    bestp = output['bestp']
    summary = side.get('summary')
    if summary is None:                       # (chains shorter than the burn-in so far)
        zmask = mu.burn_mask(output['zchain'], output['burnin'])
        summary = ms.summarise(output['posterior'][zmask])
        (output['medianp'], output['meanp'], output['stdp'],
         output['median_low_bounds'], output['median_high_bounds']) = \
            ms.statistics_from_summary(summary, bestp, inp.pstep)
    else:
        zmask = output['zmask']
    mode, hlo, hhi = ms.hpd_statistics(output['posterior'], zmask, summary, bestp, inp.pstep,
                                       kde=side.get('kde'))
    output['mode'], output['hpd_low_bounds'], output['hpd_high_bounds'] = mode, hlo, hhi
    fixed = inp.pstep == 0
    output['CRlo'] = np.where(fixed, 0.0, hlo - bestp)
    output['CRhi'] = np.where(fixed, 0.0, hhi - bestp)
    _summary_table(log, inp, output)

    if savefile is not None:
        if lead:
            grown = side.get('checkpoint')
            if grown is not None:        # the file the reports grew takes the final rows too
                grown[0].finish(savefile, output, grown[1])
            else:
                np.savez(savefile, **output)
            log.msg("\nOutput sampler files:")
            log.msg(savefile, indent=2)
        if comm is not None:
            comm.barrier()               # nobody returns before the file is complete
    if plots:
        log.warning("plots=True ignored: plotting is outside the MI355X path "
                    "(pass the output dictionary to mc3.plots)")
    if own_log:
        if lead:
            log.msg(log.logname, indent=2)
        log.close()
    return output


def _limited(*args, **kwargs):
    """mc3.sample's signature and output dictionary (`_sample` has the documentation);
    the host-side thread pools (BLAS, OpenMP) are limited to `host_threads()` for the
    duration of the call where threadpoolctl is at hand (MC3_AMD_HOST_THREADS=n overrides;
    0 leaves the pools alone)."""
    import os
    want = os.environ.get('MC3_AMD_HOST_THREADS')
    limit = host_threads() if want is None else int(want)
    if limit > 0:
        try:
            from threadpoolctl import threadpool_limits
        except ImportError:
            threadpool_limits = None
        if threadpool_limits is not None:
            with threadpool_limits(limits=limit):
                return _sample(*args, **kwargs)
    return _sample(*args, **kwargs)


# (the signature, the defaults and the documentation a caller sees are _sample's -- mc3.sample's)
sample = functools.wraps(_sample)(_limited)
