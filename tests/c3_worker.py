"""Worker for test_gpu_p2p.py::test_config_c3_*: one rank of BASELINE.json's config C3
(quadratic fit, ndata=1000, 2048 chains sharded over 8 ranks, demc) at engine level.
All ranks share device 0 (the test box has one GPU): eight processes, eight contexts of
256 chains each, eight lean_demc_kernel<3, P2P> grids co-resident (one wave per chain:
2048 waves = the chip's 8 per CU), every new row stored into all eight replicas."""
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
    out_path, nref = sys.argv[1], int(sys.argv[2])
    # key=value extras: per_rank (256), ndata (1000), sampler (demc), thinning (1), zlag (the
    # communicator's default), fx (0; 1: a fixed and a shared coefficient, priors)
    extra = dict(a.split('=') for a in sys.argv[3:])
    # exchange=rccl: the north star's collective (one in-place ncclAllGather per refresh, issued
    # by the library) instead of the peer-to-peer hand-off; devices=own: every rank on the GPU
    # of its LOCAL_RANK (tests/test_gpu_multidevice.py, boxes with two GPUs and more)
    rccl = extra.get('exchange') == 'rccl'
    if rccl and os.environ.get('MC3_TEST_RCCL_RANKS_AS_HOSTS'):
        # several ranks on ONE GPU (the development box): RCCL refuses two ranks of a host on one
        # device ("Duplicate GPU detected") -- told that every rank is a host of its own it
        # takes them as nodes of a cluster and connects them through its socket transport over
        # the loopback interface.  Not a fast path, but the collective of the north star with
        # N > 1 ranks: bootstrap, ring set-up, the in-place all-gather per refresh, RCCL's own
        # rank count -- on the box the suite runs on (tests/test_gpu_p2p.py).
        os.environ['NCCL_HOSTID'] = f"mc3hip-test-host-{os.environ.get('RANK', '0')}"
        os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
        os.environ.setdefault('NCCL_IB_DISABLE', '1')
        os.environ.setdefault('NCCL_NET', 'Socket')
        os.environ.setdefault('NCCL_SHM_DISABLE', '1')
    device = int(os.environ.get('LOCAL_RANK', '0')) if extra.get('devices') == 'own' else 0
    comm = ranks.worker_comm(use_p2p=not rccl, use_rccl=rccl)
    per_rank = int(extra.get('per_rank', 256))
    thinning = int(extra.get('thinning', 1))
    nchains = per_rank*comm.size
    prob = H.lean_shape_problem(nchains, nref, ndata=int(extra.get('ndata', 1000)),
                                sampler=extra.get('sampler', 'demc'), thinning=thinning,
                                fx=int(extra.get('fx', 0)))
    pop = np.load(out_path + '.pop.npz')
    c0, n = comm.partition(nchains)
    opts = comm.engine_options(device, thinning, int(extra['zlag']) if 'zlag' in extra else None)
    if rccl:
        assert 'p2p' not in opts, opts
    else:
        crowded = comm.size > 2 and extra.get('devices') != 'own'
        assert opts['p2p'] and opts['no_courier'] == crowded, opts
    gpu = ranks.first_gpu_call(comm, lambda: engine.DeviceChains(
        chain0=c0, nlocal=n, device=device, **opts, **prob))
    gpu.set_population(pop['Z0'], pop['lp0'])
    comm.setup(gpu)
    if rccl:
        # one launch per refresh + ONE in-place all-gather of the refresh's rows, replayed from
        # a hipGraph; RCCL itself says how many ranks the communicator spans
        assert comm.mode == 'rccl', comm.mode
        gpu.run(nref//3)
        gpu.run(nref - nref//3)
        comm.finish(gpu)
        nranks, myrank = gpu.comm_info()
        assert (nranks, myrank) == (comm.size, comm.rank), (nranks, myrank)
        plan = gpu.plan
        assert plan['comm_graph'] > 0 and plan['zlag'] == opts['zlag'], plan
        Z, lp, zc = gpu.rows()
        st = gpu.state()
        np.savez(f'{out_path}.rank{comm.rank}.npz', Z=Z, lp=lp, zc=zc, zlag=plan['zlag'],
                 numaccept=st['numaccept'], outbounds=st['outbounds'],
                 best_chisq=st['best_chisq'], X=st['X'], chisq=st['chisq'])
        print(f'rank {comm.rank}: rccl all-gather over {nranks} ranks on device {device}',
              flush=True)
        comm.barrier()
        gpu.close()
        ranks.finalize(comm)
        return
    assert comm.mode == 'p2p', comm.mode
    plan = gpu.plan
    # the hand-laid-out loops: lean_demc_kernel<3, P2P> (strict lockstep), lean_demc_lag_kernel
    # (the declared cadence), lean_snooker_kernel / lean_mrw_kernel with their couriers
    if not os.environ.get('C3_WORKER_ANYPLAN'):         # (tools/dbg_ranks.py: other kernels)
        assert plan['fast'] == 2 and plan['persistent'] > 0 and plan['zlag'] == opts['zlag'], plan
    if 'turns' in extra:                                # (kernels_rows.h, peer to peer)
        assert plan['chains_per_workgroup'] == int(extra['turns']), plan
    if 'MC3HIP_TEST_REMOTE_DELAY_NS' in os.environ:
        assert plan['remote_delay_ns'] > 0, plan
    # two segments: the second persistent launch resumes from the replicas
    gpu.run(nref//3, sync=False)
    comm.quiesce(gpu)
    gpu.run(nref - nref//3, sync=False)
    comm.quiesce(gpu)
    assert comm.mode == 'p2p' and comm.recoveries == 0
    Z, lp, zc = gpu.rows()
    st = gpu.state()
    np.savez(f'{out_path}.rank{comm.rank}.npz', Z=Z, lp=lp, zc=zc, zlag=plan['zlag'],
             numaccept=st['numaccept'], outbounds=st['outbounds'],
             best_chisq=st['best_chisq'], X=st['X'], chisq=st['chisq'])
    comm.barrier()
    gpu.close()
    ranks.finalize(comm)


if __name__ == '__main__':
    main()
