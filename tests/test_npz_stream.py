"""The checkpoint that grows in place (mc3_amd/npz_stream.py): an ordinary .npz at every
report -- np.load reads it, every member's CRC is right -- whose per-row arrays are only ever
appended to."""
import os
import zipfile
import zlib

import numpy as np
import pytest

import mc3_amd
from mc3_amd import utils as mu
from mc3_amd.npz_stream import StreamingNpz, crc32_combine
from oracle_engine import OracleChains
import helpers as H


def test_crc32_combine():
    rs = np.random.RandomState(1)
    for la, lb in ((0, 5), (1, 1), (1000, 77777), (12345, 1), (7, 1 << 20)):
        a, b = rs.bytes(la), rs.bytes(lb)
        assert crc32_combine(zlib.crc32(a), zlib.crc32(b), lb) == zlib.crc32(a + b)
    assert crc32_combine(123, 0, 0) == 123


def snapshots(nrows, rs):
    Z = rs.normal(size=(nrows[-1], 3))
    zc = (np.arange(nrows[-1]) % 7).astype(np.int32)
    lp = rs.normal(size=nrows[-1])
    for k, n in enumerate(nrows):
        out = {'posterior': Z[:n], 'zchain': zc[:n], 'log_post': lp[:n], 'chisq': -2*lp[:n],
               'bestp': np.array([1.0, 2.0, 3.0]) + k, 'pnames': np.array(['a', 'b', 'c']),
               'acceptance_rate': 25.0 + k, 'best_model': rs.normal(size=500)}
        if k > 0:
            out['zmask'] = np.arange(50, n)
        yield out


def spec(n):
    return {'posterior': (np.float64, (3,), n), 'zchain': (np.int32, (), n),
            'chisq': (np.float64, (), n), 'log_post': (np.float64, (), n),
            'zmask': (np.intp, (), n)}


def test_every_report_is_a_loadable_npz_and_costs_only_its_new_rows(tmp_path):
    path = str(tmp_path / 'run.npz')
    w = StreamingNpz(path, spec(200000))
    before = 0
    rows_before = 0
    for out in snapshots([1000, 60000, 60000, 200000], np.random.RandomState(0)):
        w.write(out)
        with np.load(path) as f:
            assert sorted(f.files) == sorted(out)
            for k, v in out.items():
                np.testing.assert_array_equal(f[k], v)
        assert zipfile.ZipFile(path).testzip() is None            # every CRC
        n = len(out['posterior'])
        new = n - rows_before
        cost = w.bytes_written - before
        # new rows (3 + 1 + 1 doubles, an int32, an index) + the small arrays, never the run so far
        assert cost <= new*(5*8 + 4 + 8) + (n - 50 if rows_before == 1000 else 0)*0 + 8*1024 \
            + (8*(n - 50) if rows_before == 1000 else 0)
        before, rows_before = w.bytes_written, n
    w.close()
    # ~2 KB of dead small members per report is all the file carries beyond its content
    want = 200000*(3*8 + 4 + 8 + 8) + 8*(200000 - 50)
    assert want < os.path.getsize(path) < want + 64*1024


def test_a_history_that_was_rewritten_is_written_again_from_the_start(tmp_path):
    path = str(tmp_path / 'run.npz')
    w = StreamingNpz(path, spec(1000))
    outs = list(snapshots([400, 900], np.random.RandomState(2)))
    w.write(outs[0])
    changed = dict(outs[1])
    changed['posterior'] = outs[1]['posterior'] + 1.0      # not a continuation of what is there
    w.write(changed)
    with np.load(path) as f:
        np.testing.assert_array_equal(f['posterior'], changed['posterior'])
        np.testing.assert_array_equal(f['log_post'], changed['log_post'])
    assert zipfile.ZipFile(path).testzip() is None
    with pytest.raises(ValueError, match='does not continue'):
        w.write(dict(outs[0]))                             # fewer rows than written
    w.close()


def test_sample_writes_a_growing_savefile_and_resumes_from_it(tmp_path):
    """sample(savefile=...) end to end on the CPU checker: the file the reports grew holds the
    final output dictionary (np.load == the returned dictionary), and a resumed run continues
    from it exactly like from np.savez's file."""
    x, data, uncert = H.tutorial_data(100)
    kw = dict(data=data, uncert=uncert, func=H.quad, params=np.array([10.0, -2.0, 0.1]),
              indparams=[x], pstep=np.array([1.0, 0.5, 0.1]),
              pmin=np.array([-10.0, -20.0, -10.0]), pmax=np.array([40.0, 20.0, 10.0]),
              sampler='demc', burnin=10, nchains=8, grtest=True, log=mu.Log(verb=0),
              engine_factory=OracleChains, seed=3)
    save = str(tmp_path / 'mcmc.npz')
    out = mc3_amd.sample(nsamples=8*200, savefile=save, **kw)
    assert zipfile.ZipFile(save).testzip() is None
    with np.load(save) as f:
        for k, v in out.items():
            np.testing.assert_array_equal(f[k], np.asarray(v), err_msg=k)
    # the same run saved the reference's way
    save2 = str(tmp_path / 'plain.npz')
    np.savez(save2, **out)
    more = mc3_amd.sample(nsamples=8*100, savefile=save, resume=True, **kw)
    more2 = mc3_amd.sample(nsamples=8*100, savefile=save2, resume=True, **kw)
    for k in ('posterior', 'zchain', 'log_post', 'chisq', 'bestp', 'acceptance_rate'):
        np.testing.assert_array_equal(more[k], more2[k], err_msg=k)
    assert len(more['posterior']) == len(out['posterior']) + 8*100
