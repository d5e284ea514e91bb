"""Worker for tools/fuzz_ranks.py: one rank of a sharded run of a randomly drawn problem
(tests/test_gpu_fuzz.py lean_problem / random_problem, keyed by a seed) at engine level, all
ranks on device 0, the peer-to-peer hand-off where the shape has one (otherwise whatever the
ranks agree on: the results must be the oracle's either way)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from mc3_amd import engine  # noqa: E402
import ranks  # noqa: E402
import fuzz_problems  # noqa: E402


def main():
    out_path, seed = sys.argv[1], int(sys.argv[2])
    comm = ranks.worker_comm(use_p2p=True, use_rccl=False)
    prob, nref, zlag = fuzz_problems.ranks_problem(seed, comm.size)
    pop = np.load(out_path + '.pop.npz')
    c0, n = comm.partition(prob['nchains'])
    opts = comm.engine_options(0, prob['thinning'], zlag)
    gpu = engine.DeviceChains(chain0=c0, nlocal=n, device=0, **opts,
                              **{k: v for k, v in prob.items() if k != 'zlag'})
    gpu.set_population(pop['Z0'], pop['lp0'])
    comm.setup(gpu)
    plan = gpu.plan
    done = 0
    for n in (nref//3, nref - nref//3):             # two segments: the second resumes
        if comm.native:
            gpu.run(n, sync=False)
            done += n
        else:                                       # (host-staged: a launch per refresh)
            for _ in range(n):
                gpu.run(1)
                comm.exchange(gpu, done)
                done += 1
        comm.quiesce(gpu)
    Z, lp, zc = gpu.rows()
    st = gpu.state()
    np.savez(f'{out_path}.rank{comm.rank}.npz', Z=Z, lp=lp, zc=zc, zlag=plan['zlag'],
             numaccept=st['numaccept'], outbounds=st['outbounds'], mode=comm.mode,
             fast=plan['fast'], cpw=plan['chains_per_workgroup'], recoveries=comm.recoveries,
             best_chisq=st['best_chisq'], X=st['X'], chisq=st['chisq'])
    comm.barrier()
    gpu.close()
    ranks.finalize(comm)


if __name__ == '__main__':
    main()
