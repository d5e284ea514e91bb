#!/usr/bin/env python3
"""The multi-GPU hand-off against an injected xGMI latency, on ONE GPU.

    python tools/delay_sweep.py [out.jsonl] [--ranks 2,4,8] [--delays 0,1000,2000,3000]
                                [--zlags 0,1,2,4,8] [--workload c2]

Runs ``bench.py --gpus N`` with every rank on device 0 (MC3HIP_BENCH_DEVICE=0) and the rows for
the other ranks' replicas held back MC3HIP_TEST_REMOTE_DELAY_NS nanoseconds (handoff.h: the
courier wave's two cursors; without a courier -- more than two ranks on a device -- the chain
wave's own queue), for the strict lockstep (zlag 0: an iteration waits for the hop) and the
declared cadence (zlag >= 1: a row has zlag iterations' time to arrive).  One JSON line per run:
ranks, delay, zlag, aggregate and per-rank rate, us per refresh, kernel.  DESIGN.md section 7's
table is made from this file (profiles/r04_delay_sweep.jsonl)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('out', nargs='?', default='gpurun_out/r04_delay_sweep.jsonl')
    ap.add_argument('--ranks', default='2,4,8')
    ap.add_argument('--delays', default='0,1000,2000,3000')
    ap.add_argument('--zlags', default='0,1,2,4,8')
    ap.add_argument('--workload', default='c2')
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--refreshes', type=int, default=16384)
    args = ap.parse_args()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, 'a') as f:
        for n in [int(v) for v in args.ranks.split(',')]:
            for zlag in [int(v) for v in args.zlags.split(',')]:
                for delay in [int(v) for v in args.delays.split(',')]:
                    env = dict(os.environ, MC3HIP_BENCH_DEVICE='0', MC3HIP_BENCH_WATCHDOG='90')
                    env.pop('MC3HIP_TEST_REMOTE_DELAY_NS', None)
                    if delay:
                        env['MC3HIP_TEST_REMOTE_DELAY_NS'] = str(delay)
                    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n),
                           '--steps', str(args.steps), '--warmup', '1', '--refreshes',
                           str(args.refreshes), '--workload', args.workload, '--zlag', str(zlag),
                           '--no-extras', '--no-cpu-baseline', '--no-e2e']
                    # (eight ranks of 256 chains fill ONE device to its last wave slot: now and
                    # then a launch of such a set-up gives up -- a rank each on its own GPU has
                    # no such neighbours -- so a failed run is tried once more and counted)
                    line, tries = None, 0
                    while line is None and tries < 2:
                        tries += 1
                        try:
                            res = subprocess.run(cmd, env=env, capture_output=True, text=True,
                                                 timeout=240)
                        except subprocess.TimeoutExpired as e:
                            res = subprocess.CompletedProcess(cmd, 1, '', f'timeout: {e}')
                        line = next((ln for ln in reversed(res.stdout.splitlines())
                                     if ln.startswith('{')), None)
                    rec = {'ranks': n, 'delay_ns': delay, 'zlag': zlag, 'workload': args.workload,
                           'tries': tries}
                    if line is None:
                        rec['error'] = (res.stderr or res.stdout)[-400:]
                    else:
                        b = json.loads(line)
                        cfg = b['config']
                        rec.update(
                            value=b['value'], per_rank=b['value']/n,
                            us_per_refresh=b['ms_per_step']*1e3/cfg['refreshes_per_step'],
                            kernel=b['roofline']['kernel'], collective=cfg['collective'],
                            zlag_run=cfg.get('zlag'), delay_run=cfg.get('remote_delay_ns'),
                            chains_per_gpu=cfg['chains_per_gpu'],
                            consistent=cfg['replicas_consistent'], recoveries=cfg['recoveries'])
                    f.write(json.dumps(rec) + '\n')
                    f.flush()
                    print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
