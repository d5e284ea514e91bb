"""Worker for test_gpu_p2p.py::test_resident_tiled_plan_sharded_*: one rank of a C5-shaped
run (design-matrix model, tiled plan) at engine level, sharded over ranks through the
peer-to-peer hand-off.  Every rank runs resident_tile_kernel<KB, P2P> over its own chains --
its tiles on chip, proposals and tile sums local -- and publishes its Z rows into every
rank's replica.  All ranks share device 0 (the test box has one GPU)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from mc3_amd import engine, parallel  # noqa: E402
import helpers as H  # noqa: E402
import ranks  # noqa: E402


def main():
    out_path, nref, per_rank, ndata, nmodel, sampler = (
        sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]),
        sys.argv[6])
    extra = dict(a.split('=') for a in sys.argv[7:])
    abort_at = int(extra['abort']) if 'abort' in extra else None
    if abort_at is not None and int(os.environ.get('RANK', '0')) == 0:
        # test hook: this rank's first chain gives up at that refresh (read at create)
        os.environ['MC3HIP_TEST_ABORT_AT'] = str(abort_at)
    comm = ranks.worker_comm(use_p2p=True, use_rccl=False)
    nchains = per_rank*comm.size
    if nmodel == 3:         # (the polynomial flavour: the tutorial's quadratic)
        prob = H.quad_problem(ndata=ndata, nchains=nchains, nsamples=nchains*nref,
                              sampler=sampler)
    else:
        prob = H.linear_problem(ndata=ndata, nmodel=nmodel, nchains=nchains,
                                nsamples=nchains*nref, sampler=sampler)
    pop = np.load(out_path + '.pop.npz')
    c0, n = comm.partition(nchains)
    trace = int(os.environ.get('C5_WORKER_TRACE', '0'))
    opts = comm.engine_options(0, 1, int(extra['zlag']) if 'zlag' in extra else None)
    gpu = engine.DeviceChains(chain0=c0, nlocal=n, device=0, trace_iters=trace, **opts, **prob)
    gpu.set_population(pop['Z0'], pop['lp0'])
    comm.setup(gpu)
    if os.environ.get('MC3HIP_NO_RESIDENT'):
        # (comparison run: three launches per iteration, host-staged exchange per refresh)
        assert comm.mode == 'host' and gpu.plan['resident'] == 0, (comm.mode, gpu.plan)
        for R in range(nref):
            gpu.run(1)
            comm.exchange(gpu, R)
    else:
        assert comm.mode == 'p2p', comm.mode
        plan = gpu.plan
        assert plan['resident'] == 1 and plan['persistent'] > 0, plan
        # two segments: the second persistent launch resumes from the replicas
        first = 0 if trace else nref//3        # (tracing: one segment, one trace row each)
        if first:
            gpu.run(first, sync=False)
            comm.quiesce(gpu)
        gpu.run(nref - first, sync=False)
        comm.quiesce(gpu)
    if os.environ.get('MC3HIP_NO_RESIDENT'):
        pass
    elif abort_at is None:
        assert comm.mode == 'p2p' and comm.recoveries == 0
    else:
        # every rank rolled back and redid the lost refreshes through the host-staged
        # exchange, one refresh per step with the three-launch plan
        assert comm.mode == 'host' and comm.recoveries == 1, (comm.mode, comm.recoveries)
        assert gpu.plan['resident'] == 0 and gpu.plan['persistent'] == 0
        print(f'rank {comm.rank}: recovered', flush=True)
    Z, lp, zc = gpu.rows()
    st = gpu.state()
    extra = {}
    if trace:
        extra['tc'], extra['tf'] = gpu.trace(trace)
    extra['zlag'] = gpu.plan['zlag']
    np.savez(f'{out_path}.rank{comm.rank}.npz', Z=Z, lp=lp, zc=zc, **extra,
             numaccept=st['numaccept'], outbounds=st['outbounds'],
             best_chisq=st['best_chisq'], X=st['X'], chisq=st['chisq'])
    comm.barrier()
    gpu.close()
    ranks.finalize(comm)


if __name__ == '__main__':
    main()
