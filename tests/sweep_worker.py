"""Worker for tests/test_gpu_instances.py: one rank of the SHARDED part of the grid -- every
case of a list (tests/instance_sweep.py ranks_grid) one after another in the same process
group, all ranks on device 0, the peer-to-peer hand-off (fuzz_worker.py does the same for one
randomly drawn problem).  argv: out prefix, grid name, first case, number of cases."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from mc3_amd import engine  # noqa: E402
import ranks  # noqa: E402
import instance_sweep as S  # noqa: E402


def main():
    out_path, grid, first, count = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    comm = ranks.worker_comm(use_p2p=True, use_rccl=False)
    cases = list(S.GRIDS[grid](comm.size))[first:first + count]
    for label, prob, opt in cases:
        comm.native, comm.mode = False, 'host'
        for key in ('MC3HIP_PERSIST_CAPACITY', 'MC3HIP_RESIDENT_MAXCOL'):
            os.environ.pop(key, None)
        os.environ.update({k: str(v) for k, v in opt.get('env', {}).items()})
        pop = np.load(f'{out_path}.{label}.pop.npz')
        c0, n = comm.partition(prob['nchains'])
        thinning = prob.get('thinning', 1)
        opts = comm.engine_options(0, thinning, prob.get('zlag', 0))
        if opt.get('no_courier'):
            opts['no_courier'] = True
        gpu = engine.DeviceChains(chain0=c0, nlocal=n, device=0, **opts,
                                  **{k: v for k, v in prob.items() if k != 'zlag'})
        gpu.set_population(pop['Z0'], pop['lp0'])
        comm.setup(gpu)
        plan = gpu.plan
        nref = gpu.nzchain
        done = 0
        for part in (nref//3, nref - nref//3):             # two segments: the second resumes
            if comm.native:
                gpu.run(part, sync=False)
                done += part
            else:                                          # (host-staged: a launch per refresh)
                for _ in range(part):
                    gpu.run(1)
                    comm.exchange(gpu, done)
                    done += 1
            comm.quiesce(gpu)
        Z, lp, zc = gpu.rows()
        st = gpu.state()
        np.savez(f'{out_path}.{label}.rank{comm.rank}.npz', Z=Z, lp=lp, zc=zc,
                 numaccept=st['numaccept'], outbounds=st['outbounds'], mode=comm.mode,
                 fast=plan['fast'], cpw=plan['chains_per_workgroup'], resident=plan['resident'],
                 tile=plan['tile'], vlanes=plan['vlanes'], tile_tree=plan['tile_tree'],
                 waves_per_chain=plan['waves_per_chain'],
                 recoveries=comm.recoveries + gpu.plan['recoveries'], kernel=gpu.last_kernel,
                 best_chisq=st['best_chisq'], X=st['X'], chisq=st['chisq'])
        comm.barrier()
        gpu.close()
        comm.barrier()
    ranks.finalize(comm)


if __name__ == '__main__':
    main()
