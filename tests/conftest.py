import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: needs a real MI355X (run by the driver with -m gpu)')
    # the launch ledgers of this session's processes (tests/ledger.py): a fresh directory the
    # rank / worker scripts inherit through the environment
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import ledger
    if not os.environ.get(ledger.ENV):
        os.environ[ledger.ENV] = tempfile.mkdtemp(prefix='mc3hip_ledger_')
        config._mc3hip_ledger_own = True


def pytest_sessionfinish(session, exitstatus):
    """A session that launched kernels leaves its instance-coverage table under gpurun_out/
    (scratch; the copy of record is profiles/rNN_instance_coverage.txt)."""
    import ledger
    try:
        total = ledger.collected()
        if total:
            text, _ = ledger.report(total)
            out = os.path.join(ROOT, 'gpurun_out')
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, 'instance_coverage.txt'), 'w') as f:
                f.write(text)
    except Exception as exc:
        print(f'instance coverage report: {exc!r}')
    if getattr(session.config, '_mc3hip_ledger_own', False):
        import shutil
        shutil.rmtree(os.environ.pop(ledger.ENV), ignore_errors=True)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
