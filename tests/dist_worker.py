"""Worker for tests/test_distributed_gloo.py: one rank of a sharded mcmc() run
(CPU engine, host-staged exchange), launched by tests/ranks.py with either communicator."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import mc3_amd  # noqa: E402
from mc3_amd import parallel, utils as mu  # noqa: E402
from oracle_engine import OracleChains  # noqa: E402
import helpers as H  # noqa: E402
import ranks  # noqa: E402


def main():
    out_path, sampler, thinning = sys.argv[1], sys.argv[2], int(sys.argv[3])
    extra = dict(a.split('=') for a in sys.argv[4:])
    zlag = {} if 'zlag' not in extra else \
        {'zlag': extra['zlag'] if extra['zlag'] == 'auto' else int(extra['zlag'])}
    comm = ranks.worker_comm(use_rccl=False)
    x, data, uncert = H.tutorial_data(100)
    np.random.seed(100 + comm.rank)          # ranks deliberately differ: rank 0's
    out = mc3_amd.sample(                    # population must be broadcast
        data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
        indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
        pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
        sampler=sampler, nsamples=8*120*thinning, burnin=10, nchains=8,
        thinning=thinning, grtest=False, log=mu.Log(verb=0),
        engine_factory=OracleChains, seed=99, comm=comm, **zlag)
    np.savez(f'{out_path}.rank{comm.rank}.npz',
             **{k: v for k, v in out.items() if k not in ('pnames', 'texnames')})
    ranks.finalize(comm)


if __name__ == '__main__':
    main()
