"""Sharding the walker ensemble over ranks (one process per GPU).

Chains are independent within an iteration; the only coupling is that proposals
read other chains' states -- for DEMC the population of the previous Z refresh,
for snooker the shared history Z (mc3/chain.py:204-217,230-232).  Rank r owns
the contiguous block of chains [r*n, (r+1)*n) and a full replica of Z; after
every refresh the new rows are exchanged so the next proposals see the whole
population:

* ``p2p`` (GPU, preferred): the persistent kernel publishes every new row straight
  into every rank's replica (peer-to-peer stores over xGMI into IPC-mapped,
  fine-grained memory) and polls its local one -- the single-GPU in-launch hand-off
  extended across GPUs; no collective, no kernel boundary between iterations
  (mc3hip_p2p_export / mc3hip_p2p_attach).
* ``rccl`` (GPU, fallback): ONE in-place ``ncclAllGather`` (RCCL over xGMI) per
  refresh issued by libmc3hip on its own stream -- mc3hip_comm_init / mc3hip_run.
* host-staged (any engine, used by the CPU tests and as a fallback): rows are read
  back, all-gathered through the control plane and installed with ``put_rows``.

The control plane -- who owns which chains, the IPC handles and the RCCL id, the
votes, the barriers at report points -- needs three primitives (all-gather of a
Python object, barrier, broadcast).  ``SocketComm`` provides them over plain TCP
sockets with nothing but the standard library (rank 0 listens, the others connect:
the rendezvous mpirun / torchrun style launchers already describe with
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); ``TorchDistComm`` is the same
communicator over an existing ``torch.distributed`` process group, for callers who
have one.  Neither is in the data path.

The random numbers are keyed by GLOBAL chain id, so a sharded run reproduces
the single-rank run bit for bit.  The reference has no counterpart: its workers
share memory inside one host (mc3/mcmc_driver.py:143-221).
"""
import os
import pickle
import socket
import struct
import tempfile
import time

import numpy as np


class SingleRank:
    rank = 0
    size = 1
    native = False
    mode = 'single'
    wants_p2p = False
    recoveries = 0

    def barrier(self):
        pass

    def quiesce(self, engine):
        pass

    def partition(self, nchains):
        return 0, nchains

    def broadcast(self, obj):
        return obj

    def all_gather(self, obj):
        return [obj]

    def setup(self, engine):
        pass

    def exchange(self, engine, refresh):
        pass

    def finish(self, engine):
        pass

    def reduce_state(self, state):
        return state

    def close(self):
        pass


class Communicator:
    """What the sharded run needs from its ranks, over three primitives a subclass
    provides: ``all_gather(obj) -> [obj of rank 0, ...]``, ``barrier()`` and
    ``broadcast(obj) -> rank 0's obj``."""

    def __init__(self, rank, size, use_rccl=True, use_p2p=True):
        self.rank = int(rank)
        self.size = int(size)
        self.use_rccl = use_rccl
        self.wants_p2p = bool(use_p2p) and self.size > 1
        self.native = False
        self.mode = 'host'
        self.recoveries = 0

    # -- primitives ---------------------------------------------------------------
    def all_gather(self, obj):
        raise NotImplementedError

    def barrier(self):
        self.all_gather(None)

    def broadcast(self, obj):
        return self.all_gather(obj if self.rank == 0 else None)[0]

    def close(self):
        pass

    # -- the protocol -------------------------------------------------------------
    def _all_ok(self, ok):
        """True iff every rank reports success."""
        return all(self.all_gather(bool(ok)))

    def partition(self, nchains):
        if nchains % self.size:
            raise ValueError(f'nchains ({nchains}) must be a multiple of the '
                             f'number of ranks ({self.size})')
        n = nchains//self.size
        return self.rank*n, n

    def setup(self, engine):
        """Choose the exchange: peer-to-peer hand-off, else RCCL inside the library,
        else host-staged.  Every step is put to the vote (``_all_ok``): either all
        ranks take it or all fall through together, so no rank is ever left waiting
        in a collective for peers that went another way."""
        if self.size == 1:
            return
        self._check_shared_device(engine)
        if self.wants_p2p and hasattr(engine, 'p2p_export'):
            handle, ok = None, False
            try:
                if engine.plan.get('persistent', 0) > 0:
                    handle = engine.p2p_export()
                    ok = True
            except Exception:
                ok = False
            if self._all_ok(ok):
                handles = self.all_gather(handle)
                # One rank at a time: concurrent hipIpcOpenMemHandle calls from several
                # processes have been seen to block each other forever (4 ranks on one
                # device, ROCm 7.2); the opens are a few milliseconds each.
                ok = True
                for turn in range(self.size):
                    if turn == self.rank:
                        try:
                            engine.p2p_attach(self.rank, handles)
                        except Exception:
                            ok = False
                    self.barrier()
                if self._all_ok(ok):
                    self.native = True
                    self.mode = 'p2p'
                    return
                if ok:
                    engine.p2p_detach()       # attached here, not everywhere: undo
                self.barrier()
        self._setup_rccl(engine)

    def _check_shared_device(self, engine):
        """Several ranks on ONE device (a test set-up; a node gives every rank its own GPU):
        the one-launch tiled plan sizes its grid for a whole device, and the grids of all the
        ranks on it must be co-resident or they wait for each other forever (until the
        hand-off gives up).  MC3HIP_RESIDENT_MAXCOL caps a rank's share; without it such a
        run is refused here, on every rank alike."""
        import os
        dev = getattr(engine, '_device', None)
        plan = getattr(engine, 'plan', None) or {}
        devices = self.all_gather((dev, bool(plan.get('resident'))))
        if dev is None:
            return
        # (the peer-to-peer instance of the headline loop has a second wave per chain, the
        # courier: the two-wave workgroups of four ranks of 256 chains would fill ONE device
        # to the last wave slot and are not all co-resident in practice -- a launch gives up
        # and the run falls back to the host-staged exchange; such a test set-up runs without
        # the courier, MC3HIP_NO_COURIER)
        if max(sum(1 for e, _ in devices if e == d) for d, _ in devices) > 2:
            os.environ.setdefault('MC3HIP_NO_COURIER', '1')
        shared = sum(1 for d, _ in devices if d == dev) > 1
        anybody = any(s and r for (d, r), s in
                      zip(devices, [sum(1 for e, _ in devices if e == d) > 1
                                    for d, _ in devices]))
        if anybody and 'MC3HIP_RESIDENT_MAXCOL' not in os.environ:
            raise RuntimeError(
                'several ranks share a device'
                + (f' (this rank: device {dev})' if shared else '')
                + ' and the one-launch tiled plan is in use: set MC3HIP_RESIDENT_MAXCOL '
                  '(columns of chains per rank) so that their grids fit the device together, '
                  'or give every rank a GPU of its own')

    def _setup_rccl(self, engine):
        """RCCL communicator inside the library, by consensus; host-staged otherwise."""
        self.native = False
        self.mode = 'host'
        if not (self.use_rccl and hasattr(engine, 'comm_init')):
            return
        from . import engine as eng
        uid, ok = None, True
        try:
            uid = eng.comm_unique_id()        # (also proves librccl loads on this rank)
        except Exception:
            ok = False
        if not self._all_ok(ok):
            return
        uid = self.broadcast(uid)
        try:
            engine.comm_init(uid, self.rank, self.size)
        except Exception:
            ok = False
        if self._all_ok(ok):
            self.native = True
            self.mode = 'rccl'
        elif ok:
            engine.comm_destroy()

    def exchange(self, engine, refresh):
        """Make the rows of ``refresh`` complete on every rank (host-staged)."""
        if self.size == 1 or self.native:
            return
        n = engine.nlocal
        row0 = engine.M0 + refresh*engine.nchains
        Z, lp, _ = engine.rows(row0 + engine.chain0, n)
        mine = np.concatenate([Z, lp[:, None]], axis=1)
        for r, a in enumerate(self.all_gather(mine)):
            if r != self.rank:
                engine.put_rows(row0 + r*n, a[:, :-1], a[:, -1])

    def quiesce(self, engine):
        """After this, the rows of every completed refresh are in the local replica
        on every rank.  With peer-to-peer stores the other ranks may still be writing
        when the local kernel has finished, and a launch may have given up on some
        rank (a wait for a peer's row timed out): the ranks vote, and if anybody
        failed everybody rolls back to the checkpoint and redoes the lost refreshes
        through RCCL or the host -- the multi-GPU counterpart of what the library
        does by itself on one GPU (mc3hip_sync)."""
        if self.size > 1 and self.mode == 'p2p':
            from ._lib import Mc3HipError, ERR_ABORTED
            ok = True
            try:
                engine.sync()
            except Mc3HipError as err:
                if err.code != ERR_ABORTED:
                    raise
                ok = False
            if not self._all_ok(ok):          # (the vote is also the barrier)
                self.recover(engine)

    def recover(self, engine):
        """Every rank: kernels have ended (sync was attempted) and the vote has been
        taken, so nobody stores into a replica any more."""
        r0, r1 = engine.p2p_rollback()
        self.recoveries += 1
        self.barrier()
        self._setup_rccl(engine)
        if self.native:
            engine.run(r1 - r0)
            engine.sync()
        else:
            for R in range(r0, r1):
                engine.run(1)
                self.exchange(engine, R)
        self.barrier()

    def finish(self, engine):
        """Complete what the per-refresh exchange leaves out (log_post rows on
        the RCCL path)."""
        if self.size > 1 and self.mode == 'rccl':
            engine.comm_gather_log_post()
        self.quiesce(engine)

    def reduce_state(self, state):
        """Gather per-chain state from every rank (chain order) and sum the
        per-parameter out-of-bounds counters."""
        if self.size == 1:
            return state
        box = self.all_gather(state)
        out = {}
        for key in state:
            if key == 'outbounds':
                out[key] = np.sum([b[key] for b in box], axis=0)
            elif key == 'refresh_done':
                out[key] = state[key]
            else:
                out[key] = np.concatenate([b[key] for b in box], axis=0)
        return out


class TorchDistComm(Communicator):
    """The communicator over an initialised ``torch.distributed`` process group (for
    callers that already have one; nothing here needs torch otherwise)."""

    def __init__(self, use_rccl=True, use_p2p=True, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        super().__init__(dist.get_rank(group), dist.get_world_size(group), use_rccl, use_p2p)

    def all_gather(self, obj):
        box = [None]*self.size
        self.dist.all_gather_object(box, obj, group=self.group)
        return box

    def barrier(self):
        self.dist.barrier(group=self.group)


# ---- the torch-free control plane -------------------------------------------------------
_HDR = struct.Struct('!Q')


def _send(sock, obj):
    blob = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(_HDR.pack(len(blob)) + blob)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError('a rank closed its control connection')
        got += k
    return bytes(buf)


def _recv(sock):
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    return pickle.loads(_recv_exact(sock, n))


def _port_file(addr, port, run_id):
    d = os.environ.get('MC3HIP_RDV_DIR') or tempfile.gettempdir()
    return os.path.join(d, f'mc3hip_rdv_{addr}_{port}_{run_id}.port')


class SocketComm(Communicator):
    """The communicator over TCP sockets, standard library only: rank 0 listens, every
    other rank keeps one connection to it; an all-gather is "everybody sends, rank 0
    answers everybody with the list" (a star: the control plane moves a few hundred
    bytes a few times per run).

    Rendezvous, ``SocketComm.from_env()``: RANK and WORLD_SIZE as every launcher sets
    them.  Where rank 0 listens:

    * ``MC3HIP_RDV_PORT`` set (our own launchers: ``bench.py --gpus N``,
      ``mc3_amd.parallel.spawn``): on MASTER_ADDR at that port;
    * otherwise (under ``torch.distributed.run``, whose agent keeps MASTER_PORT for its
      own store): on an ephemeral port that rank 0 publishes in a file named after
      MASTER_ADDR / MASTER_PORT / the launcher's run id in the temporary directory --
      one node, which is what this package shards over; the connection is confirmed by
      a token, so a stale file of an earlier run is never mistaken for this one.
    """

    def __init__(self, rank, size, addr='127.0.0.1', port=0, port_file=None, timeout=300.0,
                 use_rccl=True, use_p2p=True):
        super().__init__(rank, size, use_rccl, use_p2p)
        self.timeout = float(timeout)
        self._peers = []          # rank 0: sockets of ranks 1.., in rank order
        self._up = None           # other ranks: the socket to rank 0
        self._listener = None
        self._port_file = None
        if self.size == 1:
            return
        if self.rank == 0:
            self._serve(addr, int(port), port_file)
        else:
            self._connect(addr, int(port), port_file)

    @classmethod
    def from_env(cls, **kw):
        rank = int(os.environ.get('RANK', '0'))
        size = int(os.environ.get('WORLD_SIZE', '1'))
        addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
        if 'MC3HIP_RDV_PORT' in os.environ:
            return cls(rank, size, addr, int(os.environ['MC3HIP_RDV_PORT']), **kw)
        run_id = os.environ.get('TORCHELASTIC_RUN_ID', 'none')
        pf = _port_file(addr, os.environ.get('MASTER_PORT', '0'), run_id)
        return cls(rank, size, addr, 0, port_file=pf, **kw)

    # -- rendezvous ---------------------------------------------------------------
    def _serve(self, addr, port, port_file):
        token = os.urandom(8).hex()
        if port_file and os.path.exists(port_file):
            os.remove(port_file)                  # (left behind by a run that died)
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(self.size)
        srv.settimeout(self.timeout)
        self._listener = srv
        if port_file:
            tmp = f'{port_file}.{os.getpid()}'
            with open(tmp, 'w') as f:
                f.write(f'{srv.getsockname()[1]} {token}\n')
            os.replace(tmp, port_file)
            self._port_file = port_file
        peers = {}
        while len(peers) < self.size - 1:
            conn, _ = srv.accept()
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            conn.settimeout(self.timeout)
            hello = _recv(conn)
            if (port_file and hello.get('token') != token) or hello.get('size') != self.size \
                    or not (0 < hello.get('rank', 0) < self.size) or hello['rank'] in peers:
                _send(conn, {'ok': False})
                conn.close()
                continue
            _send(conn, {'ok': True})
            peers[hello['rank']] = conn
        self._peers = [peers[r] for r in range(1, self.size)]

    def _connect(self, addr, port, port_file):
        deadline = time.monotonic() + self.timeout
        last = None
        while time.monotonic() < deadline:
            token = None
            try:
                if port_file:
                    with open(port_file) as f:
                        p, token = f.read().split()
                    port = int(p)
                s = socket.create_connection((addr, port), timeout=self.timeout)
                s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                s.settimeout(self.timeout)
                _send(s, {'rank': self.rank, 'size': self.size, 'token': token})
                if _recv(s).get('ok'):
                    self._up = s
                    return
                s.close()
            except (OSError, ValueError, ConnectionError, EOFError) as e:   # not there yet
                last = e
            time.sleep(0.05)
        raise TimeoutError(f'rank {self.rank}: no rendezvous with rank 0 at {addr} '
                           f'({port_file or port}) within {self.timeout:.0f} s: {last}')

    # -- primitives ---------------------------------------------------------------
    def all_gather(self, obj):
        if self.size == 1:
            return [obj]
        if self.rank == 0:
            box = [obj] + [_recv(s) for s in self._peers]
            for s in self._peers:
                _send(s, box)
            return box
        _send(self._up, obj)
        return _recv(self._up)

    def close(self):
        for s in self._peers + [self._up, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._up, self._listener = [], None, None
        if self._port_file and os.path.exists(self._port_file):
            os.remove(self._port_file)
            self._port_file = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def free_port(addr='127.0.0.1'):
    with socket.socket() as s:
        s.bind((addr, 0))
        return s.getsockname()[1]


def rank_env(rank, size, port, addr='127.0.0.1', base=None):
    """The environment of rank ``rank`` of ``size`` processes on this node that meet through
    ``SocketComm.from_env()`` at ``addr:port`` (what ``bench.py --gpus N`` gives its children)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size),
               MASTER_ADDR=addr, MC3HIP_RDV_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # HIP IPC on this driver (dmabuf)
    return env
