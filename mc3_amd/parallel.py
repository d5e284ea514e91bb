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
* host-staged (any engine, used by the CPU tests with gloo and as a fallback):
  rows are read back, all-gathered through ``torch.distributed`` and installed
  with ``put_rows``.

The random numbers are keyed by GLOBAL chain id, so a sharded run reproduces
the single-rank run bit for bit.  The reference has no counterpart: its workers
share memory inside one host (mc3/mcmc_driver.py:143-221).
"""
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

    def setup(self, engine):
        pass

    def exchange(self, engine, refresh):
        pass

    def finish(self, engine):
        pass

    def reduce_state(self, state):
        return state


class TorchDistComm:
    """Communicator over an initialised ``torch.distributed`` process group."""

    def __init__(self, use_rccl=True, use_p2p=True, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.use_rccl = use_rccl
        self.wants_p2p = bool(use_p2p) and self.size > 1
        self.native = False
        self.mode = 'host'
        self.recoveries = 0

    def _all_ok(self, ok):
        """True iff every rank reports success."""
        box = [None]*self.size
        self.dist.all_gather_object(box, bool(ok), group=self.group)
        return all(box)

    def partition(self, nchains):
        if nchains % self.size:
            raise ValueError(f'nchains ({nchains}) must be a multiple of the '
                             f'number of ranks ({self.size})')
        n = nchains//self.size
        return self.rank*n, n

    def barrier(self):
        self.dist.barrier(group=self.group)

    def broadcast(self, obj):
        box = [obj if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]

    def setup(self, engine):
        """Choose the exchange: peer-to-peer hand-off, else RCCL inside the library,
        else host-staged.  Every step is put to the vote (``_all_ok``): either all
        ranks take it or all fall through together, so no rank is ever left waiting
        in a collective for peers that went another way."""
        if self.size == 1:
            return
        if self.wants_p2p and hasattr(engine, 'p2p_export'):
            handle, ok = None, False
            try:
                if engine.plan.get('persistent', 0) > 0:
                    handle = engine.p2p_export()
                    ok = True
            except Exception:
                ok = False
            if self._all_ok(ok):
                handles = [None]*self.size
                self.dist.all_gather_object(handles, handle, group=self.group)
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
                    self.dist.barrier(group=self.group)
                if self._all_ok(ok):
                    self.native = True
                    self.mode = 'p2p'
                    return
                if ok:
                    engine.p2p_detach()       # attached here, not everywhere: undo
                self.dist.barrier(group=self.group)
        self._setup_rccl(engine)

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
        """Make the rows of ``refresh`` complete on every rank."""
        if self.size == 1 or self.native:
            return
        import torch
        n = engine.nlocal
        row0 = engine.M0 + refresh*engine.nchains
        Z, lp, _ = engine.rows(row0 + engine.chain0, n)
        mine = torch.from_numpy(np.concatenate([Z, lp[:, None]], axis=1).copy())
        parts = [torch.empty_like(mine) for _ in range(self.size)]
        self.dist.all_gather(parts, mine, group=self.group)
        for r, part in enumerate(parts):
            if r == self.rank:
                continue
            a = part.numpy()
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
        self.dist.barrier(group=self.group)
        self._setup_rccl(engine)
        if self.native:
            engine.run(r1 - r0)
            engine.sync()
        else:
            for R in range(r0, r1):
                engine.run(1)
                self.exchange(engine, R)
        self.dist.barrier(group=self.group)

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
        box = [None]*self.size
        self.dist.all_gather_object(box, state, group=self.group)
        out = {}
        for key in state:
            if key == 'outbounds':
                out[key] = np.sum([b[key] for b in box], axis=0)
            elif key == 'refresh_done':
                out[key] = state[key]
            else:
                out[key] = np.concatenate([b[key] for b in box], axis=0)
        return out
