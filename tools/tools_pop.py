"""Initial population through the device (shared by the tools)."""
import numpy as np


def population(g, seed=5):
    rs = np.random.RandomState(seed)
    ifree = g.ifree
    Z0 = np.empty((0, g.nfree))
    lp0 = np.empty(0)
    while len(Z0) < g.M0:
        trial = rs.normal(g.params[ifree], g.pstep[ifree], (2*g.M0, g.nfree))
        full = g.expand_free(trial)
        ok = np.all((full >= g.pmin) & (full <= g.pmax), axis=1)
        lp = -0.5*g.eval_chisq(full[ok])
        fin = np.isfinite(lp)
        Z0 = np.concatenate([Z0, trial[ok][fin]])
        lp0 = np.concatenate([lp0, lp[fin]])
    return np.ascontiguousarray(Z0[:g.M0]), np.ascontiguousarray(lp0[:g.M0])
