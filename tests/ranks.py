"""Launching the ranks of a multi-process test, and the communicator a worker gets.

Two launchers: 'socket' -- N plain python processes that meet through
mc3_amd.parallel.SocketComm (no torch anywhere) -- and 'torchrun' --
``python -m torch.distributed.run``, the workers then use the TorchDistComm adapter over
a gloo process group."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ledger  # noqa: E402   (a rank leaves its launch ledger when it ends: tests/ledger.py)

ledger.install()


def launch_ranks(nproc, script, *args, env=None, launcher='socket', timeout=1000):
    """Run tests/<script> as `nproc` ranks on this node; returns the ranks' combined
    stdout (asserts that every rank exited with 0)."""
    from mc3_amd import parallel
    base = dict(os.environ, OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0', **(env or {}))
    for k in ('MC3HIP_RDV_PORT', 'MC3HIP_RDV_FILE', 'MC3HIP_RDV_SECRET'):
        base.pop(k, None)
    path = os.path.join(ROOT, 'tests', script)
    argv = [str(a) for a in args]
    port = parallel.free_port()
    if launcher == 'torchrun':
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
               f'--nproc-per-node={nproc}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), path] + argv
        res = subprocess.run(cmd, env=base, capture_output=True, text=True, timeout=timeout)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-6000:]
        return res.stdout
    rdv = parallel.new_rendezvous()
    procs = [subprocess.Popen([sys.executable, path] + argv,
                              env=parallel.rank_env(r, nproc, rdv, base=base),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(nproc)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f'rank {r} exited with {p.returncode}\n' + so[-2000:] + se[-6000:]
    return ''.join(so for so, _ in outs)


def worker_comm(**kw):
    """In a worker: SocketComm when launched by launch_ranks(launcher='socket'), the
    torch.distributed adapter under torchrun."""
    from mc3_amd import parallel
    if 'MC3HIP_RDV_FILE' in os.environ or 'MC3HIP_RDV_PORT' in os.environ \
            or os.environ.get('MC3_TEST_COMM') == 'socket':
        return parallel.SocketComm.from_env(**kw)
    import torch.distributed as dist
    dist.init_process_group('gloo')
    return parallel.TorchDistComm(**kw)


def dry_run():
    """MC3_TEST_DRY_RUN=1 (tests/test_multidevice_dry_run.py, a box WITHOUT a GPU): a worker goes
    as far as its first GPU call -- arguments, rendezvous, partition, the communicator's
    options, the problem, the population file -- and stops there."""
    return os.environ.get('MC3_TEST_DRY_RUN') == '1'


def first_gpu_call(comm, make):
    """`make()` is the worker's first call into the GPU (a context, or sample()).  In a dry
    run it must fail with the LIBRARY's word that there is no device -- not with a TypeError
    three lines earlier --, and the ranks leave together; otherwise its result."""
    if not dry_run():
        return make()
    from mc3_amd import _lib
    try:
        make()
    except _lib.Mc3HipError as err:
        assert err.code == _lib.ERR_HIP, err
        print(f'rank {comm.rank} of {comm.size}: dry run reached the first GPU call', flush=True)
        finalize(comm)
        sys.exit(0)
    raise AssertionError('dry run on a box that HAS a GPU: unset MC3_TEST_DRY_RUN')


def finalize(comm):
    comm.barrier()
    if hasattr(comm, 'dist'):
        comm.dist.destroy_process_group()
    else:
        comm.close()
