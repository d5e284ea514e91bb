"""Worker for test_distributed_gloo.py::test_peer_to_peer_recovery_logic_on_cpu: the
vote / roll-back / redo logic of the communicator (mc3_amd/parallel.py Communicator) without a GPU.

The engine EMULATES the peer-to-peer mode on top of the CPU checker: while attached, a rank
advances the whole ensemble itself (its replica is complete without an exchange, as if
the peers stored into it); sync() on rank 1 reports ONE aborted launch; p2p_rollback()
restores the checkpoint taken before the unconfirmed run and leaves the rank with its own
shard, so the redo and everything after go through the host-staged exchange."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import mc3_amd  # noqa: E402
from mc3_amd import parallel, utils as mu  # noqa: E402
from mc3_amd._lib import Mc3HipError, ERR_ABORTED  # noqa: E402
from oracle_engine import OracleChains  # noqa: E402
import helpers as H  # noqa: E402
import ranks  # noqa: E402

ARRAYS = ('Z', 'log_post', 'zchain', 'X', 'chisq', 'numaccept', 'outbounds', 'best_chisq',
          'best_iter', 'bestx')
FAIL_AFTER = 40          # refreshes


class EmulatedP2P(OracleChains):
    def __init__(self, chain0=0, nlocal=None, **kw):
        super().__init__(chain0=0, nlocal=None, **kw)          # the whole ensemble
        self.shard = (int(chain0), int(nlocal))
        self.chain0, self.nlocal = self.shard                   # what the hub sees
        self.plan = dict(self.plan, persistent=1)
        self.attached = False
        self.ckpt = None
        self.confirmed = True
        self.failed_once = False

    def p2p_export(self):
        return b'h'*128

    def p2p_attach(self, rank, handles):
        assert len(handles) == int(os.environ['WORLD_SIZE'])
        self.attached = True

    def run(self, nrefresh=None, sync=True):
        if self.attached and self.confirmed:
            r = self.run_
            self.ckpt = ({k: copy.deepcopy(getattr(r, k)) for k in ARRAYS},
                         int(r.S.refresh_done))
            self.confirmed = False
        return super().run(nrefresh, sync)

    def sync(self):
        if self.attached and int(os.environ['RANK']) == 1 and not self.failed_once and \
                self.refresh_done >= FAIL_AFTER:
            self.failed_once = True
            raise Mc3HipError(ERR_ABORTED, 'emulated: a wait for a peer row timed out')
        self.confirmed = True

    def p2p_rollback(self):
        r = self.run_
        saved, r0 = self.ckpt
        r1 = int(r.S.refresh_done)
        for k, v in saved.items():
            getattr(r, k)[...] = v
        r.S.refresh_done = r0
        r.P.chain0, r.P.nlocal = self.shard
        self.attached = False
        return r0, r1


def main():
    out_path = sys.argv[1]
    comm = ranks.worker_comm(use_rccl=False, use_p2p=True)
    x, data, uncert = H.tutorial_data(100)
    out = mc3_amd.sample(
        data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
        indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
        pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
        sampler='demc', nsamples=8*120, burnin=10, nchains=8, grtest=True,
        log=mu.Log(verb=0), engine_factory=EmulatedP2P, seed=99, comm=comm)
    assert comm.recoveries == 1 and comm.mode == 'host' and not comm.native, \
        (comm.recoveries, comm.mode)
    np.savez(f'{out_path}.rank{comm.rank}.npz',
             **{k: v for k, v in out.items() if k not in ('pnames', 'texnames')})
    print('recovered', flush=True)
    ranks.finalize(comm)


if __name__ == '__main__':
    main()
