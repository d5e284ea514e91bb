"""Host-side glue mirrored from the reference's utils for the sample() path:
``burn`` (mc3/utils/utils.py:260-344), default parameter names and a minimal
``Log`` whose ``error`` raises like the reference's (mc3/utils/log.py:219-240).
"""
import sys
import textwrap

import numpy as np


def default_parnames(npars):
    """'Param 1', 'Param 2', ... (mc3/utils/utils.py default_parnames)."""
    return np.array([f'Param {i+1}' for i in range(npars)])


def burn_mask(zchain, burnin, sort=True, layout=None):
    """The index array of ``burn`` alone (no row is copied).  ``layout = (first, nchains)``: the
    caller KNOWS the rows are `first` initial ones and then chains 0..nchains-1 in turn (the
    device's layout, mc3/chain.py:164,288) -- the scan that establishes it is skipped."""
    if layout is not None:
        first, nchains = layout
        niter = (len(zchain) - first)//nchains
        it = np.arange(min(burnin, niter), niter, dtype=np.int64)
        if sort:
            zmask = np.empty((nchains, len(it)), np.int64)
            np.add(np.arange(nchains, dtype=np.int64)[:, None] + first, (it*nchains)[None, :],
                   out=zmask)
        else:
            zmask = np.empty((len(it), nchains), np.int64)
            np.add((it*nchains)[:, None] + first, np.arange(nchains, dtype=np.int64)[None, :],
                   out=zmask)
        return zmask.reshape(-1)
    return burn(Z=np.empty((len(zchain), 0)), zchain=zchain, burnin=burnin, sort=sort)[2]


def burn(Zdict=None, burnin=None, Z=None, zchain=None, sort=True):
    """Remove the first ``burnin`` samples of every chain.

    Returns (posterior, zchain, zmask) with zmask the indices that turn Z into
    the burned posterior, optionally grouped by chain (mc3/utils/utils.py:260-344).
    """
    if Zdict is None and (Z is None or zchain is None or burnin is None):
        raise ValueError(
            'Need to input either Zdict or all three of burnin, Z, and zchain')
    if Zdict is not None:
        Z = Zdict['posterior']
        zchain = Zdict['zchain']
        if burnin is None:
            burnin = Zdict['burnin']
    zchain = np.asarray(zchain)
    # Fast path for the layout the device produces (and mrw/demc in the reference,
    # mc3/chain.py:164,288): optional initial rows marked -1, then rows in
    # repeating chain order 0..n-1.  Same result as the general scan below.
    nvalid0 = int(np.argmax(zchain >= 0)) if np.any(zchain >= 0) else len(zchain)
    body = zchain[nvalid0:]
    nchains_r = int(body.max()) + 1 if len(body) else 0
    if nchains_r > 0 and len(body) % nchains_r == 0 and np.all(zchain[:nvalid0] < 0) and \
            np.array_equal(body.reshape(-1, nchains_r),
                           np.broadcast_to(np.arange(nchains_r, dtype=body.dtype),
                                           (len(body)//nchains_r, nchains_r))):
        niter = len(body)//nchains_r
        it = np.arange(min(burnin, niter), niter)
        if sort:
            zmask = (nvalid0 + it[None, :]*nchains_r + np.arange(nchains_r)[:, None]).ravel()
        else:
            zmask = (nvalid0 + it[:, None]*nchains_r + np.arange(nchains_r)[None, :]).ravel()
        return Z[zmask], zchain[zmask], zmask
    keep = np.zeros(len(zchain), bool)
    for c in range(int(np.amax(zchain)) + 1):
        keep[np.flatnonzero(zchain == c)[burnin:]] = True
    if sort:
        order = np.argsort(zchain, kind='stable')
        zmask = order[keep[order]]
    else:
        zmask = np.flatnonzero(keep)
    return Z[zmask], zchain[zmask], zmask


class Log:
    """Screen (and optional file) logger with the reference's call surface:
    msg / warning / error / progressbar / close."""

    def __init__(self, logname=None, verb=2, append=False, width=70):
        self.logname = logname
        self.verb = verb
        self.width = width
        self.warnings = []
        self.sep = 70*':'
        self.file = None
        if logname is not None:
            self.file = open(logname, 'a' if append else 'w')

    def write(self, text):
        print(text, flush=True)
        if self.file is not None and not self.file.closed:
            self.file.write(text + '\n')
            self.file.flush()

    def wrap(self, text, indent=0, si=None):
        lines = []
        for line in str(text).splitlines():
            lines.append(textwrap.fill(
                line, self.width, initial_indent=' '*indent,
                subsequent_indent=' '*(indent if si is None else si),
                break_long_words=False, break_on_hyphens=False) if line else '')
        return '\n'.join(lines)

    def msg(self, message, verb=1, indent=0, si=None, noprint=False, width=None):
        if verb > self.verb:
            return
        if width is not None:
            old, self.width = self.width, width
            text = self.wrap(message, indent, si)
            self.width = old
        else:
            text = self.wrap(message, indent, si)
        if noprint:
            return text
        self.write(text)

    def warning(self, message):
        if self.verb < 1:
            return
        text = self.wrap(message, indent=4)
        self.warnings.append(text)
        self.write(f'\n{self.sep}\n  Warning:\n{text}\n{self.sep}\n')

    def error(self, error_message, exception=ValueError, tracklev=None):
        if self.file is not None and not self.file.closed:
            self.file.write(f'\n{self.sep}\n{self.wrap(error_message)}\n{self.sep}')
            self.close()
        raise exception(error_message)

    def progressbar(self, frac):
        bar = int(np.clip(10*frac, 0, 10))
        self.msg(f"\n[{':'*bar:<10s}] {100*frac:5.1f}% completed", verb=1)

    def close(self):
        if self.file is not None and not self.file.closed:
            self.file.close()
