"""oracle/kernels.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes wrappers around oracle_kernels.c with the call signatures of the
reference's Python wrappers (mc3/stats/stats.py:97-284), plus ``log_prior``
(stats.py:287-392) restated in NumPy.  ``load_ref()`` imports the reference's
own compiled C kernels from oracle/_ref when they have been built.
"""
import ctypes
import importlib.machinery
import importlib.util
import os
import subprocess
import sysconfig

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, 'liboracle.so')
    if force or not os.path.exists(so):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'liboracle.so'],
                              stdout=subprocess.DEVNULL)
    subprocess.call(['make', '-C', _HERE, 'ref'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.oracle_priors.restype = ctypes.c_double
        L.oracle_chisq.restype = ctypes.c_double
        L.oracle_dwt_chisq.restype = ctypes.c_double
        L.oracle_daub4.restype = ctypes.c_int
        L.oracle_residuals.restype = None
        for name in ('ls_det_exp', 'ls_det_exp2', 'ls_det_log'):
            getattr(L, name).restype = ctypes.c_double
            getattr(L, name).argtypes = [ctypes.c_double]
        L.ls_det_pow_half.restype = ctypes.c_double
        L.ls_det_pow_half.argtypes = [ctypes.c_double, ctypes.c_int]
        L.ls_det_sincos2pi.restype = None
        L.ls_det_sincos2pi.argtypes = [ctypes.c_double, _dp, _dp]
        _LIB = L
    return _LIB


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _prior_args(params, priors, priorlow, priorup):
    """The Gaussian-prior mask of stats.py:211-216."""
    if params is None or priors is None or priorlow is None or priorup is None:
        return None, None, None, 0, ()
    params = np.asarray(params, float)
    priors = np.asarray(priors, float)
    priorlow = np.asarray(priorlow, float)
    priorup = np.asarray(priorup, float)
    iprior = (priorlow > 0) & (priorup > 0)
    off, poff = _c((params - priors)[iprior])
    lo, plo = _c(priorlow[iprior])
    up, pup = _c(priorup[iprior])
    return poff, plo, pup, len(off), (off, lo, up)


def chisq(model, data, uncert, params=None, priors=None, priorlow=None,
          priorup=None):
    """mc3.stats.chisq (stats.py:150-216 -> _chisq.c:111-140)."""
    model, pm = _c(model)
    data, pd = _c(data)
    uncert, pu = _c(uncert)
    poff, plo, pup, npri, _keep = _prior_args(params, priors, priorlow, priorup)
    return lib().oracle_chisq(pm, pd, pu, ctypes.c_int(len(model)),
                              poff, plo, pup, ctypes.c_int(npri))


def residuals(model, data, uncert, params=None, priors=None, priorlow=None,
              priorup=None):
    """mc3.stats.residuals (stats.py:97-147 -> _chisq.c:37-79)."""
    model, pm = _c(model)
    data, pd = _c(data)
    uncert, pu = _c(uncert)
    poff, plo, pup, npri, _keep = _prior_args(params, priors, priorlow, priorup)
    out = np.empty(len(model) + npri)
    lib().oracle_residuals(pm, pd, pu, ctypes.c_int(len(model)),
                           poff, plo, pup, ctypes.c_int(npri),
                           out.ctypes.data_as(_dp))
    return out


def dwt_chisq(model, data, params, priors=None, priorlow=None, priorup=None):
    """mc3.stats.dwt_chisq (stats.py:219-284 -> _dwt.c:56-119)."""
    if len(params) < 3:
        raise ValueError('Wavelet chisq should have at least three parameters')
    model, pm = _c(model)
    data, pd = _c(data)
    params = np.ascontiguousarray(params, float)
    n = len(data)
    if n < 2 or (n & (n - 1)) != 0:
        raise ValueError(
            'oracle: the wavelet likelihood is defined only for power-of-two '
            'ndata (the reference reads unwritten scratch otherwise)')
    poff, plo, pup, npri, _keep = _prior_args(params, priors, priorlow, priorup)
    wres = np.empty(n)
    tmp = np.empty(n)
    return lib().oracle_dwt_chisq(
        params.ctypes.data_as(_dp), ctypes.c_int(len(params)), pm, pd,
        ctypes.c_int(n), poff, plo, pup, ctypes.c_int(npri),
        wres.ctypes.data_as(_dp), tmp.ctypes.data_as(_dp))


def dwt_daub4(array, inverse=False):
    """mc3.stats.dwt_daub4 (stats.py -> _dwt.c:154-186)."""
    array, pa = _c(array)
    n = len(array)
    m = int(np.ceil(np.log2(n))) if n > 1 else 0
    nw = 1 << m
    out = np.empty(nw)
    tmp = np.empty(nw)
    isign = -1 if inverse else 1
    lib().oracle_daub4(pa, ctypes.c_int(n), ctypes.c_int(isign),
                       out.ctypes.data_as(_dp), tmp.ctypes.data_as(_dp))
    return out


def log_prior(posterior, prior, priorlow, priorup, pstep):
    """mc3.stats.log_prior (stats.py:287-392): -0.5*sum of squared prior
    residuals for Gaussian priors, 2 log(p) entries for priorlow<0."""
    posterior = np.atleast_2d(np.asarray(posterior, float))
    prior = np.asarray(prior, float)
    priorlow = np.asarray(priorlow, float)
    priorup = np.asarray(priorup, float)
    pstep = np.asarray(pstep, float)
    ifree = np.where(pstep > 0)[0]
    delta = posterior - prior[ifree]
    for i, j in enumerate(ifree):
        if priorlow[j] > 0 and priorup[j] > 0:
            col = delta[:, i]
            neg = col < 0
            pos = col > 0
            col[neg] /= priorlow[j]
            col[pos] /= priorup[j]
        elif priorlow[j] < 0:
            delta[:, i] = 2.0*np.log(posterior[:, i])
        else:
            delta[:, i] = 0.0
    logp = -0.5*np.sum(delta**2, axis=1)
    if np.size(logp) == 1:
        return logp[0]
    return logp


def load_ref():
    """The reference's own _chisq/_dwt C modules from oracle/_ref, or None."""
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    mods = []
    for name in ('_chisq', '_dwt'):
        path = os.path.join(_HERE, '_ref', name + ext)
        if not os.path.exists(path):
            return None
        loader = importlib.machinery.ExtensionFileLoader(name, path)
        spec = importlib.util.spec_from_file_location(name, path, loader=loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        mods.append(mod)
    return tuple(mods)
