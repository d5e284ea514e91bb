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

The random numbers are keyed by GLOBAL chain id, so a sharded run reproduces the
single-rank run bit for bit -- by DEFAULT: like the reference, whose semantics do not
depend on ``ncpu`` (mc3/chain.py:221-232), a run's results do not depend on the number of
ranks.  The strict lockstep (``zlag = 0``) is the default for every world size; the relaxed
cadence that hides the xGMI hop (``zlag >= 1``: the other chains are seen that many refreshes
late) is OPT-IN -- ``zlag=`` / ``zlag='auto'`` (CADENCE_ZLAG refreshes' worth)
or MC3HIP_ZLAG -- is written to the output dictionary, the savefile and the log, and a
resumed run refuses a savefile of another cadence (mcmc()).  N ranks with a given zlag equal
ONE rank with the same zlag.  The reference has no counterpart: its workers share memory
inside one host (mc3/mcmc_driver.py:143-221).
"""
import hashlib
import hmac
import os
import pickle
import socket
import struct
import tempfile
import time

import numpy as np


# Refreshes (of thinning 1) a row may take to reach the other ranks in the declared cadence of
# a sharded run (opt-in: zlag='auto'): an iteration of the headline loop is ~0.6 us, so four of
# them cover an xGMI hop of up to ~2.5 us (profiles/r04_delay_sweep.jsonl: where the rate
# starts to fall per zlag)
CADENCE_ZLAG = 4


def resolve_zlag(zlag, thinning=1):
    """The cadence a run asks for: an integer as given; 'auto' = CADENCE_ZLAG iterations' worth
    of refreshes; None = MC3HIP_ZLAG if set (an integer or 'auto'), else 0 -- the strict
    lockstep, whatever the number of ranks."""
    if zlag is None:
        zlag = os.environ.get('MC3HIP_ZLAG', 0)
    if isinstance(zlag, str):
        zlag = zlag.strip().lower()
        if zlag == 'auto':
            return -(-CADENCE_ZLAG//max(1, int(thinning)))
    zlag = int(zlag)
    if zlag < 0:
        raise ValueError(f'zlag must be >= 0 or "auto" (got {zlag})')
    return zlag


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

    def engine_options(self, device=0, thinning=1, zlag=None):
        zlag = resolve_zlag(zlag, thinning)
        return {'zlag': zlag} if zlag else {}

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
        self.crowded = False

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

    def engine_options(self, device=0, thinning=1, zlag=None):
        """What this communicator asks of the engine it is about to be given (keywords of
        DeviceChains), decided BEFORE the engine exists -- the library sizes its persistent
        grid from the very kernel instance these select:

        * ``zlag``: the cadence (include/mc3hip.h): the other chains are seen ``zlag`` refreshes
          late, so that a row has that long to cross xGMI.  Default 0 -- the strict lockstep,
          the SAME random process as one GPU (every iteration then pays the hop); 'auto' =
          CADENCE_ZLAG iterations' worth of refreshes; MC3HIP_ZLAG when nothing is given
          (``resolve_zlag``); rank 0's word counts;
        * ``p2p``: fine-grained, IPC-shareable Z (the peer-to-peer hand-off is wanted);
        * ``no_courier``: more than two ranks share this rank's device (a test set-up)."""
        zlag = resolve_zlag(zlag, thinning)
        if self.size == 1:
            return {'zlag': zlag} if zlag else {}
        opts = {'zlag': int(self.broadcast(int(zlag)))}
        if self.wants_p2p:
            devices = self.all_gather(int(device))
            self.crowded = max(devices.count(d) for d in devices) > 2
            opts.update(p2p=True, no_courier=self.crowded)
        return opts

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
        dev = getattr(engine, '_device', None)
        plan = getattr(engine, 'plan', None) or {}
        # the persistent kernels wait for each other's rows, so the grids of ALL the ranks on a
        # device must be co-resident: each library checks its own grid against the device, the
        # sum is checked here (seen in round 4: lag-loop builds that held one wave per SIMD
        # instead of two passed the per-rank check and deadlocked beyond 1024 workgroups)
        grid = -(-int(getattr(engine, 'nlocal', 0))//max(1, int(plan.get('chains_per_workgroup', 1))))
        load = self.all_gather((dev, grid if plan.get('persistent') and not plan.get('resident')
                                else 0, int(plan.get('capacity', 0))))
        for d in {d for d, _, _ in load if d is not None}:
            on = [(g, c) for e, g, c in load if e == d]
            caps = [c for g, c in on if g and c]
            # (MC3HIP_PERSIST_CAPACITY: a test hook that makes the library pretend fewer fit)
            if len(on) > 1 and caps and sum(g for g, _ in on) > min(caps) \
                    and not os.environ.get('MC3HIP_PERSIST_CAPACITY'):
                raise RuntimeError(
                    f'the ranks on device {d} launch {sum(g for g, _ in on)} workgroups that must '
                    f'be co-resident; the device holds {min(caps)} of this kernel: use fewer '
                    'chains per rank, or give every rank a GPU of its own')
        devices = self.all_gather((dev, bool(plan.get('resident'))))
        if dev is None:
            return
        # (the peer-to-peer instances of the headline loops have a second wave per chain, the
        # courier: the two-wave workgroups of four ranks of 256 chains would fill ONE device
        # to the last wave slot and are not all co-resident in practice -- a launch gives up
        # and the run falls back to the host-staged exchange; such a test set-up runs without
        # the courier, which the engine must be CREATED for: engine_options() says so)
        if self.wants_p2p and max(sum(1 for e, _ in devices if e == d) for d, _ in devices) > 2 \
                and not getattr(engine, 'no_courier', True) \
                and not os.environ.get('MC3HIP_NO_COURIER'):
            raise RuntimeError(
                'more than two ranks share a device: create the engine with the keywords of '
                'Communicator.engine_options() (no_courier=True)')
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
                if os.environ.get('MC3HIP_VERBOSE'):
                    print(f'[rank {self.rank}] {err}', flush=True)
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
_MAX_FRAME = 1 << 30


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
    """A pickled frame -- ONLY ever called on a connection that has passed the handshake
    below (unpickling runs code: nothing a stranger sent may reach it)."""
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if n > _MAX_FRAME:
        raise ConnectionError(f'control frame of {n} bytes')
    return pickle.loads(_recv_exact(sock, n))


# The handshake is fixed-size binary, no pickle: rank 0 sends a 16-byte challenge; the rank
# answers with MAGIC | rank | size | HMAC-SHA256(secret, challenge | rank | size); rank 0 checks
# it in constant time and answers with HMAC(secret, challenge | b'ok') -- so both sides prove
# they hold the run's secret (MC3HIP_RDV_SECRET, made by the launcher, or the token of the
# port file, which the reading side accepts only from a 0600 file of its own uid in a 0700
# directory of its own: _read_port_file / _private_dir) before a single frame is unpickled.
_MAGIC = b'MC3H'
_HELLO = struct.Struct('!4sII32s')


def _mac(secret, *parts):
    return hmac.new(secret, b''.join(parts), hashlib.sha256).digest()


def _handshake_server(conn, secret, size, taken):
    """The rank behind an accepted connection, or None (a stranger, garbage, a duplicate)."""
    challenge = os.urandom(16)
    conn.sendall(challenge)
    magic, rank, nranks, mac = _HELLO.unpack(_recv_exact(conn, _HELLO.size))
    want = _mac(secret, challenge, struct.pack('!II', rank, nranks))
    if magic != _MAGIC or not hmac.compare_digest(mac, want) or nranks != size \
            or not (0 < rank < size) or rank in taken:
        return None
    conn.sendall(_mac(secret, challenge, b'ok'))
    return rank


def _handshake_client(sock, secret, rank, size):
    challenge = _recv_exact(sock, 16)
    sock.sendall(_HELLO.pack(_MAGIC, rank, size,
                             _mac(secret, challenge, struct.pack('!II', rank, size))))
    answer = _recv_exact(sock, 32)
    return hmac.compare_digest(answer, _mac(secret, challenge, b'ok'))


def _is_loopback(addr):
    return addr in ('localhost', '::1') or addr.startswith('127.')


def _private_dir():
    """Where rendezvous files live: a directory only this user can enter -- MC3HIP_RDV_DIR if
    given, else <tmp>/mc3hip_rdv_<uid> created 0700.  Either way it is CHECKED (ours, a real
    directory, no group / other bits): in a world-writable sticky /tmp a predictable file name
    could be planted by another user before rank 0 writes it, and the ranks would then prove
    themselves to that user's server and unpickle its frames."""
    d = os.environ.get('MC3HIP_RDV_DIR')
    if not d:
        d = os.path.join(tempfile.gettempdir(), f'mc3hip_rdv_{os.getuid()}')
        try:
            os.mkdir(d, 0o700)
        except FileExistsError:
            pass
    st = os.lstat(d)
    import stat as _stat
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid():
        raise PermissionError(f'SocketComm: rendezvous directory {d} is not a directory of '
                              f'uid {os.getuid()}: refusing to meet there')
    if st.st_mode & 0o077:
        if os.environ.get('MC3HIP_RDV_DIR'):
            raise PermissionError(f'SocketComm: MC3HIP_RDV_DIR={d} is open to group / others '
                                  f'(mode {st.st_mode & 0o777:o}): make it 0700')
        os.chmod(d, 0o700)
    return d


def _port_file(addr, port, run_id):
    return os.path.join(_private_dir(), f'mc3hip_rdv_{addr}_{port}_{run_id}.port')


def _read_port_file(path):
    """(port, token) of the file rank 0 published -- trusted ONLY if it is a regular file of
    this very user that nobody else can read or write (no symlink followed).  A file that
    fails the check is an attack or a misconfiguration, never "not there yet": PermissionError,
    which the rendezvous does not retry."""
    import errno
    import stat as _stat
    try:
        fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
    except OSError as e:
        if e.errno == errno.ELOOP:
            raise PermissionError(f'SocketComm: {path} is a symbolic link: refusing it') from e
        raise
    try:
        st = os.fstat(fd)
        if not _stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or st.st_mode & 0o077:
            raise PermissionError(
                f'SocketComm: {path} is not a private file of uid {os.getuid()} (owner '
                f'{st.st_uid}, mode {st.st_mode & 0o7777:o}): refusing its port and token')
        with os.fdopen(os.dup(fd)) as f:
            p, token = f.read().split()
    finally:
        os.close(fd)
    return int(p), token


class SocketComm(Communicator):
    """The communicator over TCP sockets, standard library only: rank 0 listens, every
    other rank keeps one connection to it; an all-gather is "everybody sends, rank 0
    answers everybody with the list" (a star: the control plane moves a few hundred
    bytes a few times per run).

    Rendezvous, ``SocketComm.from_env()``: RANK and WORLD_SIZE as every launcher sets
    them.  Where rank 0 listens:

    * ``MC3HIP_RDV_FILE`` set (our own launchers: ``bench.py --gpus N``, tests/ranks.py, through
      ``rank_env``): on MASTER_ADDR at a port of rank 0's own choosing, published in that file
      (nobody probes for a free port and lets go of it again);
    * ``MC3HIP_RDV_PORT`` set: on MASTER_ADDR at that port (``MC3HIP_RDV_SECRET`` required);
    * otherwise (under ``torch.distributed.run``, whose agent keeps MASTER_PORT for its
      own store): on an ephemeral port that rank 0 publishes in a file named after
      MASTER_ADDR / MASTER_PORT / the launcher's run id in a directory only this user can
      enter (``_private_dir``: <tmp>/mc3hip_rdv_<uid>, 0700, ownership and mode checked) --
      one node, which is what this package shards over.

    Every connection proves, in a fixed-size binary handshake (no pickle), that it holds the
    run's secret -- ``MC3HIP_RDV_SECRET`` or the token in the port file -- before a frame of it
    is unpickled; strangers and garbage are dropped and the rendezvous goes on.  The port file
    is trusted only if it is a regular file of this uid with mode 0600 (opened O_NOFOLLOW and
    fstat'ed, ``_read_port_file``); a foreign or group / world-accessible file ends the run
    with PermissionError on the reading ranks, and rank 0 refuses to replace a file it does
    not own.
    Only loopback addresses unless ``MC3HIP_RDV_ALLOW_REMOTE=1``.
    """

    def __init__(self, rank, size, addr='127.0.0.1', port=0, port_file=None, timeout=300.0,
                 use_rccl=True, use_p2p=True, secret=None, comm_timeout=None):
        super().__init__(rank, size, use_rccl, use_p2p)
        # `timeout` bounds the rendezvous only; between collectives the ranks may be busy for
        # as long as a run takes (a checkpoint, a library rebuild on one rank): a dead peer
        # closes its socket, a hung one is the launcher's watchdog's business
        self.timeout = float(timeout)
        self.comm_timeout = float(os.environ.get('MC3HIP_COMM_TIMEOUT', 86400.0)
                                  if comm_timeout is None else comm_timeout)
        self._secret = secret.encode() if isinstance(secret, str) else secret
        self._peers = []          # rank 0: sockets of ranks 1.., in rank order
        self._up = None           # other ranks: the socket to rank 0
        self._listener = None
        self._port_file = None
        if self.size == 1:
            return
        if not _is_loopback(addr) and not os.environ.get('MC3HIP_RDV_ALLOW_REMOTE'):
            raise ValueError(
                f'SocketComm: {addr} is not a loopback address; this package shards over the '
                'GPUs of ONE node.  Set MC3HIP_RDV_ALLOW_REMOTE=1 to listen there all the same')
        if self._secret is None and not port_file:
            raise ValueError('SocketComm: a fixed port needs the run\'s secret '
                             '(MC3HIP_RDV_SECRET; mc3_amd.parallel.rank_env makes one)')
        if self.rank == 0:
            self._serve(addr, int(port), port_file)
        else:
            self._connect(addr, int(port), port_file)

    @classmethod
    def from_env(cls, **kw):
        rank = int(os.environ.get('RANK', '0'))
        size = int(os.environ.get('WORLD_SIZE', '1'))
        addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
        secret = os.environ.get('MC3HIP_RDV_SECRET')
        if 'MC3HIP_RDV_FILE' in os.environ:       # (our launchers: rank 0 picks the port)
            return cls(rank, size, addr, 0, port_file=os.environ['MC3HIP_RDV_FILE'],
                       secret=secret, **kw)
        if 'MC3HIP_RDV_PORT' in os.environ:
            return cls(rank, size, addr, int(os.environ['MC3HIP_RDV_PORT']), secret=secret, **kw)
        run_id = os.environ.get('TORCHELASTIC_RUN_ID', 'none')
        pf = _port_file(addr, os.environ.get('MASTER_PORT', '0'), run_id)
        return cls(rank, size, addr, 0, port_file=pf, secret=secret, **kw)

    # -- rendezvous ---------------------------------------------------------------
    def _serve(self, addr, port, port_file):
        token = os.urandom(16).hex()
        if port_file and os.path.lexists(port_file):
            # left behind by a run of ours that died -- anybody else's file (or one we cannot
            # remove: a sticky directory) ends the run HERE, loudly: the other ranks would
            # refuse it anyway (_read_port_file)
            st = os.lstat(port_file)
            if st.st_uid != os.getuid():
                raise PermissionError(f'SocketComm: {port_file} exists and belongs to uid '
                                      f'{st.st_uid}: refusing to publish the rendezvous there')
            os.remove(port_file)
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(self.size)
        srv.settimeout(self.timeout)
        self._listener = srv
        # the secret of this run: the launcher's, or the token only this user can read
        secret = self._secret if self._secret is not None else token.encode()
        if port_file:
            tmp = f'{port_file}.{os.getpid()}'
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
            with os.fdopen(fd, 'w') as f:
                f.write(f'{srv.getsockname()[1]} {token}\n')
            os.replace(tmp, port_file)
            self._port_file = port_file
        peers = {}
        deadline = time.monotonic() + self.timeout
        while len(peers) < self.size - 1:
            if time.monotonic() > deadline:
                raise TimeoutError(f'rank 0: {self.size - 1 - len(peers)} ranks did not show up '
                                   f'within {self.timeout:.0f} s')
            conn, _ = srv.accept()
            # a connection that sends garbage, the wrong secret or nothing at all is dropped
            # and the rendezvous goes on: only its own ranks can end it
            try:
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(min(self.timeout, 20.0))
                rank = _handshake_server(conn, secret, self.size, peers)
            except (OSError, ConnectionError, struct.error):
                rank = None
            if rank is None:
                try:
                    conn.close()
                except OSError:
                    pass
                continue
            conn.settimeout(self.comm_timeout)
            peers[rank] = conn
        self._peers = [peers[r] for r in range(1, self.size)]

    def _connect(self, addr, port, port_file):
        deadline = time.monotonic() + self.timeout
        last = None
        while time.monotonic() < deadline:
            secret = self._secret
            s = None
            try:
                if port_file:
                    port, token = _read_port_file(port_file)
                    if secret is None:
                        secret = token.encode()
                s = socket.create_connection((addr, port), timeout=self.timeout)
                s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                s.settimeout(self.timeout)
                if _handshake_client(s, secret, self.rank, self.size):
                    s.settimeout(self.comm_timeout)
                    self._up = s
                    return
                last = ConnectionError('rank 0 did not prove the secret of this run')
                s.close()
            except PermissionError:
                raise                             # (a foreign port file: never retried)
            except (OSError, ValueError, ConnectionError, EOFError) as e:   # not there yet
                last = e
                if s is not None:
                    s.close()
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
    """A port that was free a moment ago (for launchers that must name one up front, like
    ``torch.distributed.run --master-port``; our own ranks meet through ``rank_env``'s file)."""
    with socket.socket() as s:
        s.bind((addr, 0))
        return s.getsockname()[1]


def new_rendezvous():
    """(file, secret) for one launch of ranks on this node: rank 0 publishes its port in the
    file, everybody proves the secret."""
    return (os.path.join(_private_dir(), f'mc3hip_rdv_{os.getpid()}_{os.urandom(6).hex()}.port'),
            os.urandom(16).hex())


def rank_env(rank, size, rendezvous, addr='127.0.0.1', base=None):
    """The environment of rank ``rank`` of ``size`` processes on this node that meet through
    ``SocketComm.from_env()`` (what ``bench.py --gpus N`` gives its children).  ``rendezvous``
    is ``new_rendezvous()``'s pair, the same for all ranks of a launch -- or a port number (the
    secret then comes from ``base`` / the environment, MC3HIP_RDV_SECRET)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size), MASTER_ADDR=addr)
    for k in ('MC3HIP_RDV_PORT', 'MC3HIP_RDV_FILE'):
        env.pop(k, None)
    if isinstance(rendezvous, (tuple, list)):
        env.update(MC3HIP_RDV_FILE=rendezvous[0], MC3HIP_RDV_SECRET=rendezvous[1])
    else:
        env.update(MC3HIP_RDV_PORT=str(rendezvous))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # HIP IPC on this driver (dmabuf)
    return env
