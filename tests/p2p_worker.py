"""Worker for tests/test_gpu_p2p.py: one rank of a sharded mcmc() run on the GPU
with the peer-to-peer hand-off.  All ranks share device 0 (the test box has one
GPU): two processes, two contexts, two persistent kernels resident at once, each
publishing its rows into both replicas through IPC-mapped memory."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import mc3_amd  # noqa: E402
from mc3_amd import parallel, utils as mu  # noqa: E402
import helpers as H  # noqa: E402
import ranks  # noqa: E402


def main():
    out_path, sampler, thinning, niter = (sys.argv[1], sys.argv[2], int(sys.argv[3]),
                                          int(sys.argv[4]))
    extra = dict(a.split('=') for a in sys.argv[5:])
    abort_at = int(extra['abort']) if 'abort' in extra else None
    nchains = int(extra.get('nchains', 16))
    if abort_at is not None and int(os.environ.get('RANK', '0')) == 0:
        # test hook: this rank's first chain gives up at that refresh (read at create)
        os.environ['MC3HIP_TEST_ABORT_AT'] = str(abort_at)
    comm = ranks.worker_comm(use_p2p=True, use_rccl=False)
    x, data, uncert = H.tutorial_data(100)
    np.random.seed(100 + comm.rank)
    out = ranks.first_gpu_call(comm, lambda: mc3_amd.sample(
        data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
        indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
        pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
        sampler=sampler, nsamples=nchains*niter*thinning, burnin=10, nchains=nchains,
        thinning=thinning, grtest=True, log=mu.Log(verb=0), seed=99, comm=comm,
        device=int(os.environ.get('LOCAL_RANK', '0'))
        if os.environ.get('P2P_WORKER_DEVICES') == 'own' else 0, **({'zlag': extra['zlag'] if extra['zlag'] == 'auto' else int(extra['zlag'])}
                     if 'zlag' in extra else {})))
    if abort_at is None:
        assert comm.mode == 'p2p' and comm.recoveries == 0, (comm.mode, comm.recoveries)
    else:
        assert comm.mode == 'host' and comm.recoveries == 1, (comm.mode, comm.recoveries)
        print(f'rank {comm.rank}: recovered', flush=True)
    np.savez(f'{out_path}.rank{comm.rank}.npz',
             **{k: v for k, v in out.items() if k not in ('pnames', 'texnames')})
    ranks.finalize(comm)


if __name__ == '__main__':
    main()
