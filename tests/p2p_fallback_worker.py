"""Worker for test_peer_to_peer_falls_back_together: rank 1's engine refuses the
peer-to-peer export, so BOTH ranks must end up on the host-staged exchange."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from mc3_amd import parallel  # noqa: E402
from oracle_engine import OracleChains  # noqa: E402
import helpers as H  # noqa: E402
import ranks  # noqa: E402


class Flaky(OracleChains):
    """Pretends to offer the peer-to-peer API; fails on rank 1."""
    plan = {'persistent': 1}

    def p2p_export(self):
        if int(os.environ['RANK']) == 1:
            raise RuntimeError('no IPC here')
        return b'x'*128

    def p2p_attach(self, rank, handles):
        raise AssertionError('attach must not be reached when an export failed')


def main():
    comm = ranks.worker_comm(use_p2p=True, use_rccl=False)
    prob = H.quad_problem(nchains=8, nsamples=8*20, sampler='demc')
    c0, n = comm.partition(8)
    eng = Flaky(chain0=c0, nlocal=n, **prob)
    eng.plan = {'persistent': 1}
    comm.setup(eng)
    assert comm.mode == 'host' and not comm.native, comm.mode
    ranks.finalize(comm)
    print('fallback ok')


if __name__ == '__main__':
    main()
