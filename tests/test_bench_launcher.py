"""bench.py --gpus N without a GPU: the self-launching parent (N rank processes, rank 0's
line relayed, non-zero exit when a rank dies), the torch-free rendezvous of the ranks and
the shape of the line, driven with the --dry-run engine (which computes nothing and says
so in `data`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def run(cmd, env=None, timeout=300):
    e = dict(os.environ, **(env or {}))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MC3HIP_RDV_PORT', 'MC3HIP_RDV_FILE',
              'MC3HIP_RDV_SECRET'):
        e.pop(k, None)
    return subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize('n,workload', [(3, 'c2'), (4, 'c3')])
def test_plain_process_starts_its_ranks(n, workload):
    res = run([sys.executable, BENCH, '--gpus', str(n), '--steps', '2', '--warmup', '1',
               '--dry-run', '--workload', workload])
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == n and out['steps'] == 2 and out['warmup'] == 1
    assert out['data'].startswith('dry-run')
    assert out['scaling'] == ('weak' if workload == 'c2' else 'strong')
    assert out['config']['nchains'] == (256*n if workload == 'c2' else 2048)
    assert out['config']['chains_per_gpu'] == (256 if workload == 'c2' else 2048//n)
    assert out['config']['replicas_consistent'] is True
    assert 'SocketComm' in out['config']['control_plane']
    for key in ('metric', 'value', 'unit', 'ms_per_step', 'higher_is_better', 'vs_baseline',
                'dtype', 'roofline'):
        assert key in out
    src = open(BENCH).read()
    assert 'import torch' not in src and 'from torch' not in src


def test_a_rank_that_dies_fails_the_run():
    res = run([sys.executable, BENCH, '--gpus', '3', '--steps', '1', '--warmup', '0',
               '--dry-run'], env={'MC3HIP_BENCH_DRY_FAIL_RANK': '1',
                                  'MC3HIP_BENCH_WATCHDOG': '20'}, timeout=200)
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert 'rank 1 exited with 3' in res.stderr


def test_ranks_started_by_torchrun_meet_without_torch():
    """The driver's documented way of starting N > 1: the launcher's processes are the ranks;
    they find each other through the port file (the launcher keeps MASTER_PORT)."""
    pytest.importorskip('torch')
    from mc3_amd import parallel
    res = run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
               '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
               str(parallel.free_port()), BENCH, '--gpus', '2', '--steps', '2', '--warmup', '1',
               '--dry-run'], timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['replicas_consistent'] is True


def test_single_rank_needs_no_rendezvous():
    res = run([sys.executable, BENCH, '--steps', '1', '--warmup', '0', '--dry-run'])
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][0])
    assert out['n_gpus'] == 1 and out['config']['control_plane'] is None
