"""The N>1 path on CPU: two ranks each own half of the chains, exchange the new Z rows
after every refresh, and must reproduce the single-rank run bit for bit (the draws are
keyed by global chain id).  Run on both control planes: mc3_amd.parallel.SocketComm
(plain TCP, no torch) and the TorchDistComm adapter over a gloo process group."""
import os
import subprocess
import sys

import numpy as np
import pytest

import mc3_amd
from mc3_amd import parallel, utils as mu
from oracle_engine import OracleChains
import helpers as H
from ranks import launch_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the first `import torch` in a fresh container can take minutes in every worker
pytestmark = pytest.mark.timeout(1200)


@pytest.mark.parametrize('sampler,thinning,launcher,zlag', [
    ('demc', 1, 'socket', None), ('snooker', 2, 'socket', None), ('demc', 1, 'torchrun', None),
    ('demc', 1, 'socket', 'auto'), ('snooker', 2, 'socket', 'auto'), ('demc', 1, 'socket', 2)])
def test_two_ranks_reproduce_single_rank(tmp_path, sampler, thinning, launcher, zlag):
    """a sharded run equals the single-rank run bit for bit -- with the DEFAULTS
    (zlag None: the strict lockstep on any number of ranks, one random process like the
    reference's, whose semantics do not depend on ncpu) and in the opt-in cadence (N ranks with
    a zlag equal ONE rank with the same zlag); the cadence is on record in the output."""
    out_path = str(tmp_path / 'shard')
    launch_ranks(2, 'dist_worker.py', out_path, sampler, thinning,
                 *([] if zlag is None else [f'zlag={zlag}']), launcher=launcher)

    x, data, uncert = H.tutorial_data(100)
    np.random.seed(100)
    single = mc3_amd.sample(
        data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
        indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
        pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
        sampler=sampler, nsamples=8*120*thinning, burnin=10, nchains=8,
        thinning=thinning, grtest=False, log=mu.Log(verb=0),
        engine_factory=OracleChains, seed=99, **({} if zlag is None else {'zlag': zlag}))
    want = 0 if zlag is None else -(-parallel.CADENCE_ZLAG//thinning) if zlag == 'auto' else zlag
    assert single['zlag'] == want
    for rank in (0, 1):
        got = np.load(f'{out_path}.rank{rank}.npz')
        for key in ('posterior', 'zchain', 'log_post', 'chisq', 'bestp',
                    'acceptance_rate', 'best_log_post', 'medianp', 'stdp', 'zlag'):
            np.testing.assert_array_equal(got[key], single[key], err_msg=key)


def test_communicator_mode_selection_single_process():
    """Mode selection logic without a process group: a single rank never sets up
    an exchange; the partition is the whole ensemble."""
    from mc3_amd import parallel
    comm = parallel.SingleRank()
    assert comm.partition(12) == (0, 12)
    assert comm.mode == 'single' and not comm.wants_p2p and not comm.native
    assert comm.broadcast({'a': 1}) == {'a': 1}
    state = {'numaccept': np.arange(3)}
    assert comm.reduce_state(state) is state


def test_peer_to_peer_falls_back_together():
    """If any rank cannot export its replica, every rank drops the peer-to-peer
    mode (no rank may end up alone in it)."""
    assert launch_ranks(2, 'p2p_fallback_worker.py').count('fallback ok') == 2


@pytest.mark.parametrize('launcher', ['socket', 'torchrun'])
def test_peer_to_peer_recovery_logic_on_cpu(tmp_path, launcher):
    """One rank reports an aborted peer-to-peer launch: the ranks vote at the next report
    point, BOTH roll back to their checkpoints and redo the lost refreshes through the
    host-staged exchange; the run ends identical to the undisturbed single-rank run
    (Communicator.quiesce / recover; the GPU form is tests/test_gpu_p2p.py)."""
    out_path = str(tmp_path / 'recover')
    assert launch_ranks(2, 'p2p_recover_worker.py', out_path, launcher=launcher) \
        .count('recovered') == 2
    x, data, uncert = H.tutorial_data(100)
    single = mc3_amd.sample(
        data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
        indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
        pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
        sampler='demc', nsamples=8*120, burnin=10, nchains=8, grtest=True,
        log=mu.Log(verb=0), engine_factory=OracleChains, seed=99)
    for rank in (0, 1):
        got = np.load(f'{out_path}.rank{rank}.npz')
        for key in ('posterior', 'zchain', 'log_post', 'chisq', 'bestp',
                    'acceptance_rate', 'best_log_post', 'medianp', 'stdp'):
            np.testing.assert_array_equal(got[key], single[key], err_msg=key)


def _socket_ranks(n, env_of_rank, body):
    """n python processes running `body` (source text) with a SocketComm named comm."""
    code = ('import os, sys\nsys.path.insert(0, %r)\nfrom mc3_amd import parallel\n'
            'comm = parallel.SocketComm.from_env(timeout=60)\n' % ROOT) + body + '\ncomm.close()\n'
    procs = [subprocess.Popen([sys.executable, '-c', code], env=env_of_rank(r),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(n)]
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
    return [so for so, _ in outs]


BODY = (
    'box = comm.all_gather({"r": comm.rank, "blob": bytes(1000*comm.rank)})\n'
    'assert [b["r"] for b in box] == list(range(comm.size))\n'
    'assert [len(b["blob"]) for b in box] == [1000*r for r in range(comm.size)]\n'
    'assert comm.broadcast(("from", comm.rank)) == ("from", 0)\n'
    'comm.barrier()\n'
    'assert comm._all_ok(True) and not comm._all_ok(comm.rank != 1)\n'
    'import numpy as np\n'
    'parts = comm.all_gather(np.full((3, 2), float(comm.rank)))\n'
    'assert all(np.all(p == r) for r, p in enumerate(parts))\n'
    'print("rank", comm.rank, "of", comm.size, "ok")\n')


def test_socket_comm_primitives_launcher_rendezvous():
    """SocketComm as bench.py --gpus N and tests/ranks.py hand it to their children
    (rank_env: rank 0 picks the port and publishes it, everybody proves the launch's secret):
    all-gather, broadcast, barrier, votes."""
    from mc3_amd import parallel
    rdv = parallel.new_rendezvous()
    outs = _socket_ranks(3, lambda r: parallel.rank_env(r, 3, rdv), BODY)
    assert sorted(o.strip() for o in outs) == [f'rank {r} of 3 ok' for r in range(3)]
    assert not os.path.exists(rdv[0])


def test_socket_comm_fixed_port_needs_and_checks_the_secret():
    """MC3HIP_RDV_PORT: a fixed port works with MC3HIP_RDV_SECRET and is refused without;
    while the ranks meet, strangers knock -- one closes at once, one sends garbage, one sends a
    well-formed hello signed with ANOTHER secret, one sends a pickle (which must never be
    loaded: it would create a file) -- and the rendezvous goes on undisturbed."""
    import pickle
    import socket
    import struct
    import threading
    import time
    from mc3_amd import parallel
    port = parallel.free_port()
    with pytest.raises(ValueError, match='secret'):
        parallel.SocketComm(1, 2, port=port)
    base = dict(os.environ, MC3HIP_RDV_SECRET='s3cret-of-this-run')
    marker = f'/tmp/mc3hip_pwned_{os.getpid()}'

    class Boom:
        def __reduce__(self):
            return (open, (marker, 'w'))

    def strangers():
        blob = pickle.dumps(Boom())
        payloads = [b'', os.urandom(200),
                    struct.pack('!4sII32s', b'MC3H', 1, 3, bytes(32)),
                    struct.pack('!Q', len(blob)) + blob]
        for _ in range(200):
            for pl in payloads:
                try:
                    with socket.create_connection(('127.0.0.1', port), timeout=1) as s:
                        s.settimeout(1)
                        s.recv(16)
                        s.sendall(pl)
                        time.sleep(0.01)
                except OSError:
                    pass
            if done.is_set():
                return

    done = threading.Event()
    t = threading.Thread(target=strangers, daemon=True)

    def env(r):
        if r == 0:
            t.start() if not t.is_alive() and not done.is_set() else None
        return parallel.rank_env(r, 3, port, base=base)
    try:
        outs = _socket_ranks(3, env, 'import time\ntime.sleep(0.5 if comm.rank else 0)\n' + BODY)
    finally:
        done.set()
    assert sorted(o.strip() for o in outs) == [f'rank {r} of 3 ok' for r in range(3)]
    assert not os.path.exists(marker), 'a frame of an unauthenticated connection was unpickled'


def test_socket_comm_refuses_addresses_that_are_not_loopback(monkeypatch):
    from mc3_amd import parallel
    monkeypatch.delenv('MC3HIP_RDV_ALLOW_REMOTE', raising=False)
    with pytest.raises(ValueError, match='loopback'):
        parallel.SocketComm(0, 2, addr='10.1.2.3', port=5555, secret='x')


def test_socket_comm_rendezvous_through_a_port_file(tmp_path):
    """Under a launcher that keeps MASTER_PORT for itself (torch.distributed.run) rank 0
    listens on an ephemeral port and publishes it in a file; a stale file of a dead run
    with the same name is ignored (the token does not match / nobody listens)."""
    from mc3_amd import parallel
    stale = parallel._port_file('127.0.0.1', '29999', 'run7')

    def env(r):
        e = dict(os.environ, RANK=str(r), WORLD_SIZE='4', MASTER_ADDR='127.0.0.1',
                 MASTER_PORT='29999', TORCHELASTIC_RUN_ID='run7', MC3HIP_RDV_DIR=str(tmp_path))
        for k in ('MC3HIP_RDV_PORT', 'MC3HIP_RDV_FILE', 'MC3HIP_RDV_SECRET'):
            e.pop(k, None)
        return e
    stale = os.path.join(str(tmp_path), os.path.basename(stale))
    with open(os.open(stale, os.O_WRONLY | os.O_CREAT, 0o600), 'w') as f:
        f.write('1 deadbeef\n')           # (port 1: nobody listens there; ours, 0600: trusted)
    outs = _socket_ranks(4, env, BODY)
    assert sorted(o.strip() for o in outs) == [f'rank {r} of 4 ok' for r in range(4)]
    assert not os.path.exists(stale), 'rank 0 removes its port file when it closes'


def test_port_file_is_trusted_only_if_private_and_ours(tmp_path, monkeypatch):
    """the token of the port file is the handshake secret under torchrun, so the
    READING side must check who wrote it -- a regular file of this uid, mode 0600, no symlink,
    in a directory nobody else can enter; anything else is PermissionError (not retried), and
    rank 0 refuses to publish over a file it does not own."""
    from mc3_amd import parallel
    d = tmp_path / 'rdv'
    d.mkdir(mode=0o700)
    monkeypatch.setenv('MC3HIP_RDV_DIR', str(d))
    good = parallel._port_file('127.0.0.1', '29998', 'runA')
    assert os.path.dirname(good) == str(d)
    with open(os.open(good, os.O_WRONLY | os.O_CREAT, 0o600), 'w') as f:
        f.write('4242 cafe\n')
    assert parallel._read_port_file(good) == (4242, 'cafe')
    os.chmod(good, 0o644)                       # readable by others: the token is no secret
    with pytest.raises(PermissionError, match='private'):
        parallel._read_port_file(good)
    os.chmod(good, 0o600)
    link = str(d / 'link.port')
    os.symlink(good, link)
    with pytest.raises(PermissionError, match='symbolic'):
        parallel._read_port_file(link)
    # a rank that meets such a file gives up at once (no retry until the timeout)
    os.chmod(good, 0o666)
    import time
    t0 = time.monotonic()
    with pytest.raises(PermissionError):
        parallel.SocketComm(1, 2, port_file=good, timeout=30.0)
    assert time.monotonic() - t0 < 5.0
    # the directory itself: open to others -> refused
    os.chmod(str(d), 0o755)
    with pytest.raises(PermissionError, match='0700'):
        parallel._port_file('127.0.0.1', '1', 'x')
    os.chmod(str(d), 0o700)
    # the default directory is created private
    monkeypatch.delenv('MC3HIP_RDV_DIR')
    monkeypatch.setattr(parallel.tempfile, 'gettempdir', lambda: str(tmp_path))
    pd = parallel._private_dir()
    assert os.stat(pd).st_mode & 0o777 == 0o700 and os.stat(pd).st_uid == os.getuid()
    if os.getuid() == 0:
        # somebody else's file under the expected name: readers refuse it, rank 0 refuses to
        # replace it
        monkeypatch.setenv('MC3HIP_RDV_DIR', str(d))
        os.chmod(good, 0o600)
        os.chown(good, 65534, -1)
        with pytest.raises(PermissionError, match='private'):
            parallel._read_port_file(good)
        with pytest.raises(PermissionError, match='belongs to uid'):
            parallel.SocketComm(0, 2, port_file=good, timeout=5.0)
        os.chown(str(d), 65534, -1)
        with pytest.raises(PermissionError, match='not a directory of uid'):
            parallel._private_dir()
        os.chown(str(d), 0, -1)
