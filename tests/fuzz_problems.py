"""The sharded problems of tools/fuzz_ranks.py, a pure function of (seed, number of ranks) so
that the parent (the oracle) and every rank build the same one."""
import numpy as np

import test_gpu_fuzz as F


def ranks_setup(seed):
    """(number of ranks, injected delay in ns, chains per wave to force or 0) of a seed."""
    rs = np.random.RandomState(770000 + seed)
    nranks = int(rs.choice([2, 2, 3, 4]))
    delay = int(rs.choice([0, 1500, 3000]))
    turns = int(rs.choice([0, 0, 2, 3, 4]))
    return nranks, delay, turns


def ranks_problem(seed, nranks):
    """A lean-family problem (polynomial of 2..4 coefficients; plain, fixed / shared / priors,
    two tiles or thinned) with an ensemble that splits evenly over the ranks; (problem,
    refreshes, zlag)."""
    rs = np.random.RandomState(880000 + seed)
    prob, flavour = F.lean_problem(seed)
    per_rank = int(rs.randint(4, 28))
    nref = int(rs.randint(12, 50))
    zlag = int(rs.choice([0, 1, 2, 4, 4]))
    prob.update(nchains=per_rank*nranks, nsamples=per_rank*nranks*nref*prob['thinning'],
                zlag=zlag)
    return prob, nref, zlag
