"""The model family the device path evaluates, and how a user callable is
mapped onto it.

The reference calls an arbitrary Python ``func(params, *indparams)`` once per
proposal (mc3/chain.py:317-319).  The HIP kernels evaluate two families
in-kernel: polynomials in one abscissa and general linear models
``A @ params``.  A model is either DECLARED with the callables below (they are
ordinary mc3 model functions and also carry the device description) or
RECOGNISED: ``lower()`` probes a plain callable for linearity in its parameters
and refuses, loudly, anything else.
"""
import numpy as np


class Polynomial:
    """func(params, x) = sum_k params[k] * x**k,  k < ncoef."""
    family = 'poly'

    def __init__(self, ncoef):
        self.ncoef = int(ncoef)

    def __call__(self, params, x):
        x = np.asarray(x, float)
        model = np.full(x.shape, float(params[self.ncoef - 1]))
        for k in range(self.ncoef - 2, -1, -1):
            model = model*x + params[k]
        return model


class Linear:
    """func(params, A) = params @ A with A of shape [ncoef, ndata]."""
    family = 'linear'

    def __init__(self, ncoef):
        self.ncoef = int(ncoef)

    def __call__(self, params, A):
        return np.asarray(params[:self.ncoef], float) @ np.asarray(A, float)


class Lowered:
    """Device description of a model: family, coefficient count, the
    independent array the kernel reads and a constant offset folded into the
    data (model = offset + family(params[:ncoef]))."""

    def __init__(self, family, ncoef, indep, offset=None):
        self.family = family
        self.ncoef = ncoef
        self.indep = np.ascontiguousarray(indep, float)
        self.offset = offset


# Polynomials of five or more coefficients over many points are handed to the device as a
# design matrix of the powers of x: as such they run on the one-launch tiled plan (matrix
# cores, 2..15 columns); as polynomials they would take the three-launch tiled plan -- the
# degree-specialised one-launch kernels stop at four coefficients.  (The powers are formed
# here with x**k; the reference's model is the user's callable, whose rounding is its own.)
POLY_AS_LINEAR_NDATA = 8192


def _polynomial(ncoef, x, ndata, wlike, offset=None):
    x = np.asarray(x, float)
    if not wlike and 5 <= ncoef <= 15 and ndata >= POLY_AS_LINEAR_NDATA and x.shape == (ndata,):
        return Lowered('linear', ncoef, np.array([x**k for k in range(ncoef)]), offset)
    return Lowered('poly', ncoef, x, offset)


def lower(func, params, indparams, indparams_dict, ndata, wlike, pstep,
          rtol=1e-9):
    """Map ``func`` onto the device family or raise ValueError."""
    params = np.asarray(params, float)
    nmodel = len(params) - 3 if wlike else len(params)
    family = getattr(func, 'family', None)
    if family == 'poly':
        if func.ncoef != nmodel:
            raise ValueError(f'Polynomial({func.ncoef}) needs {func.ncoef} model '
                             f'parameters, got {nmodel}')
        return _polynomial(nmodel, indparams[0], ndata, wlike)
    if family == 'linear':
        A = np.asarray(indparams[0], float)
        if A.shape != (nmodel, ndata):
            raise ValueError(f'Linear model needs A of shape ({nmodel}, {ndata})')
        return Lowered('linear', nmodel, A)

    # a plain callable: accept it iff it is affine in params[:nmodel]
    def f(p):
        return np.asarray(func(p, *indparams, **indparams_dict), float)

    base = np.zeros(nmodel)
    with np.errstate(all='ignore'):
        f0 = f(base)
        cols = np.empty((nmodel, ndata))
        for k in range(nmodel):
            e = base.copy()
            e[k] = 1.0
            cols[k] = f(e) - f0
        rs = np.random.RandomState(12345)
        scale = np.where(np.asarray(pstep)[:nmodel] != 0,
                         np.abs(np.asarray(pstep)[:nmodel]), 1.0)
        ok = np.all(np.isfinite(f0)) and np.all(np.isfinite(cols))
        for _ in range(4):
            if not ok:
                break
            p = params[:nmodel] + rs.normal(0, 3, nmodel)*scale
            want = f(p)
            got = f0 + p @ cols
            tol = rtol*(np.abs(want) + np.abs(p) @ np.abs(cols) + np.abs(f0) + 1e-300)
            ok = bool(np.all(np.abs(got - want) <= tol))
    if not ok:
        raise ValueError(
            "'func' is not linear in its parameters: the MI355X path evaluates "
            "polynomial / linear models in-kernel (mc3_amd.models.Polynomial, "
            "mc3_amd.models.Linear, or any callable affine in params); use the "
            "reference mc3.sample() for other models")
    offset = f0 if np.any(f0 != 0.0) else None
    # polynomial in a single abscissa?  (cheaper: the kernel reads x only)
    if len(indparams) == 1 and np.shape(indparams[0]) == (ndata,):
        x = np.asarray(indparams[0], float)
        powers = np.array([x**k for k in range(nmodel)])
        if np.allclose(cols, powers, rtol=1e-13, atol=0.0):
            return _polynomial(nmodel, x, ndata, wlike, offset)
    return Lowered('linear', nmodel, cols, offset)
