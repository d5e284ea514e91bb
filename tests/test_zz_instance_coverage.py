"""Runs LAST (the file name): every kernel instance libmc3hip.so carries must have been
launched by the GPU tests that ran before it -- tests which compare what the launch wrote with
the oracle (tests/oracle_engine.py, oracle/) or with the reference's own numbers.  The library
counts its launches itself (include/mc3hip.h mc3hip_debug_launch_counts); the processes of the
session leave their ledgers in $MC3HIP_LEDGER_DIR (tests/ledger.py, tests/conftest.py).

An instance nobody launched is either dead weight -- delete it from the library -- or a hole
in the suite -- add the case that reaches it.  tests/instance_allowlist.txt names the few that
cannot run on the one-GPU box the suite runs on, with the reason.

The path under test: mc3/chain.py:183-299 of the reference (one code path there; here one
instance per shape of problem)."""
import os
import re

import pytest

import ledger

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ALLOW = os.path.join(HERE, 'instance_allowlist.txt')


def allowlist():
    """[(regular expression, reason)] from tests/instance_allowlist.txt ('pattern | reason')."""
    rules = []
    if os.path.exists(ALLOW):
        with open(ALLOW) as f:
            for line in f:
                line = line.strip()
                if line and not line.startswith('#'):
                    pat, _, why = line.partition('|')
                    assert why.strip(), f'allowlist entry without a reason: {line}'
                    rules.append((re.compile(pat.strip()), why.strip()))
    return rules


def whole_suite_ran(session):
    """Only a session that ran every GPU test module can say an instance was NEVER launched."""
    if session.config.option.keyword or session.config.getoption('-x') and session.testsfailed:
        return False
    ran = {os.path.basename(str(item.fspath)) for item in session.items}
    want = {f for f in os.listdir(HERE) if f.startswith('test_gpu_') and f.endswith('.py')}
    return want <= ran


def test_every_kernel_instance_of_the_library_was_launched(request):
    if not whole_suite_ran(request.session):
        pytest.skip('needs the whole GPU suite in one session (python -m pytest tests -m gpu)')
    text, never = ledger.report()
    rules = allowlist()
    excused = [(n, why) for n in never for pat, why in rules if pat.search(n)]
    missing = [n for n in never if not any(pat.search(n) for pat, _ in rules)]
    text += '\nexcused by tests/instance_allowlist.txt:\n' + \
        ''.join(f'  {n}    [{why}]\n' for n, why in excused)
    out = os.path.join(os.path.dirname(HERE), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'instance_coverage.txt'), 'w') as f:
        f.write(text)
    assert not missing, (f'{len(missing)} kernel instances of libmc3hip.so were never launched '
                         f'by the suite:\n  ' + '\n  '.join(missing[:40]))
