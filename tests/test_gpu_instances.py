"""Every kernel instance the library carries, launched beside the oracle: the grid of
tests/instance_sweep.py walked deliberately (degree x sampler x what the parameters carry x
thinning x tiles x chains per wave x cadence x ranks), exact comparison for each.  What the
launch ledger then still lists as never launched (tests/test_zz_instance_coverage.py) has no
business in the library.  Reference path: mc3/chain.py:183-299."""
import itertools
import os

import numpy as np
import pytest

import helpers as H
import instance_sweep as S
from test_gpu_parity import assert_same, run_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from mc3_amd import engine
    return engine


class env:
    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def capacity_for(prob, cpw):
    """MC3HIP_PERSIST_CAPACITY (test plumbing: 'pretend fewer workgroups fit') that makes the
    library advance `cpw` chains per workgroup."""
    return -(-prob['nchains']//cpw)


def check(eng, prob, cpw=1, trace=False, want_cpw=True):
    kw = {} if cpw == 1 else dict(MC3HIP_PERSIST_CAPACITY=capacity_for(prob, cpw))
    with env(**kw):
        orc, gpu = run_pair(eng, prob, trace=trace)
        plan = gpu.plan
        kernel = gpu.last_kernel
        assert plan['recoveries'] == 0, plan
        assert_same(orc, gpu, trace=trace)
        gpu.close()
    if want_cpw and plan['persistent'] > 0:
        assert plan['chains_per_workgroup'] == cpw, (plan, kernel)
    return plan, kernel


LEAN = {label: (prob, cpw) for label, prob, cpw in
        itertools.chain(S.one_gpu_lean_grid(), S.one_gpu_big_ensembles_grid())}
LAG = {label: (prob, cpw) for label, prob, cpw in S.one_gpu_lag_grid()}


@pytest.mark.parametrize('label', sorted(LEAN))
def test_lean_family_on_one_gpu(eng, label):
    """Polynomials of 2..4 coefficients: the hand-laid-out loops, one chain per wave and two to
    four side by side / in turns, every sampler and flavour."""
    prob, cpw = LEAN[label]
    check(eng, prob, cpw)


@pytest.mark.parametrize('label', sorted(LAG))
def test_declared_cadence_on_one_gpu(eng, label):
    prob, cpw = LAG[label]
    check(eng, prob, cpw)


GENERAL = {label: (prob, opt) for label, prob, opt in S.general_grid()}
WAVELET = {label: (prob, opt) for label, prob, opt in S.wavelet_grid()}
TILED = {label: (prob, opt) for label, prob, opt in S.tiled_grid()}


def check_general(eng, prob, opt):
    """the general chain: also its chisq / model evaluation kernels for the same geometry"""
    cpw = opt.get('cpw', 1)
    kw = {} if cpw == 1 else dict(MC3HIP_PERSIST_CAPACITY=capacity_for(prob, cpw))
    gpu_kw = {k: opt[k] for k in ('jit', 'no_persist') if opt.get(k) is not None and
              (k != 'no_persist' or opt[k])}
    with env(**kw):
        orc, gpu = run_pair(eng, prob, trace=bool(opt.get('trace')), **gpu_kw)
        plan = gpu.plan
        assert plan['recoveries'] == 0 and plan['fast'] in (0, 3), (plan, gpu.last_kernel)
        if opt.get('no_persist'):
            assert plan['persistent'] == 0, plan
        elif cpw > 1:
            assert plan['chains_per_workgroup'] == cpw, (plan, gpu.last_kernel)
        assert_same(orc, gpu, trace=bool(opt.get('trace')))
        pts = np.array([orc.params, orc.params*1.01])
        np.testing.assert_array_equal(gpu.eval_chisq(pts), orc.eval_chisq(pts))
        np.testing.assert_array_equal(gpu.eval_model_values(orc.params), orc.eval_model(orc.params))
        gpu.close()


@pytest.mark.parametrize('label', sorted(GENERAL))
def test_general_chain_on_one_gpu(eng, label):
    check_general(eng, *GENERAL[label])


@pytest.mark.parametrize('label', sorted(WAVELET))
def test_wavelet_family_on_one_gpu(eng, label):
    check_general(eng, *WAVELET[label])


@pytest.mark.parametrize('label', sorted(TILED))
def test_tiled_plans_on_the_vector_unit(eng, label):
    prob, opt = TILED[label]
    gpu_kw = dict(jit=False, **({'no_persist': True} if opt.get('no_persist') else {}))
    orc, gpu = run_pair(eng, prob, trace=False, **gpu_kw)
    plan = gpu.plan
    assert plan['tile'] > 0 and plan['recoveries'] == 0, (plan, gpu.last_kernel)
    assert_same(orc, gpu, trace=False)
    gpu.close()


# ---- sharded over two ranks on the one GPU: the peer-to-peer flavours -------------------------
def run_ranks_grid(grid, nranks, tmp_path, chunk=400):
    """Every case of a grid through tests/sweep_worker.py (ONE process group for all of them),
    every rank's replica against the oracle's run of the whole ensemble."""
    import ranks
    cases = list(S.GRIDS[grid](nranks))
    out = str(tmp_path/'s')
    pops = {}
    for label, prob, opt in cases:
        orc = H.make_oracle(prob)
        Z0, lp0 = H.initial_population(orc)
        np.savez(f'{out}.{label}.pop.npz', Z0=Z0, lp0=lp0)
        pops[label] = (Z0, lp0)
    for first in range(0, len(cases), chunk):
        ranks.launch_ranks(nranks, 'sweep_worker.py', out, grid, first, chunk, timeout=900)
    oracles = {}
    for label, prob, opt in cases:
        # (the oracle adds up in the order the library reports: the plan of rank 0)
        got = np.load(f'{out}.{label}.rank0.npz')
        plan = {k: int(got[k]) for k in ('tile', 'vlanes', 'tile_tree', 'waves_per_chain')}
        orc = H.make_oracle(prob, plan=plan)
        orc.set_population(*pops[label])
        orc.run(nthreads=8)
        oracles[label] = orc
    kernels = {}
    for label, prob, opt in cases:
        orc = oracles[label]
        per = prob['nchains']//nranks
        for r in range(nranks):
            got = np.load(f'{out}.{label}.rank{r}.npz')
            s = slice(per*r, per*(r + 1))
            msg = f'{label} rank {r} ({got["kernel"]})'
            assert int(got['recoveries']) == 0, msg
            np.testing.assert_array_equal(got['Z'], orc.Z, err_msg=msg)
            np.testing.assert_array_equal(got['lp'], orc.log_post, err_msg=msg)
            np.testing.assert_array_equal(got['numaccept'], orc.numaccept[s], err_msg=msg)
            np.testing.assert_array_equal(got['X'], orc.X[s], err_msg=msg)
            np.testing.assert_array_equal(got['chisq'], orc.chisq[s], err_msg=msg)
            np.testing.assert_array_equal(got['best_chisq'], orc.best_chisq[s], err_msg=msg)
            kernels[label] = (str(got['mode']), str(got['kernel']))
    return kernels


def test_lean_family_sharded_over_two_ranks(tmp_path):
    kernels = run_ranks_grid('lean', 2, tmp_path)
    # (strict DEMC with a prior / a fixed coefficient and several chains per wave has no
    # peer-to-peer loop: those cases go through the host -- everything else hands over in-launch)
    host = {k for k, v in kernels.items() if v[0] != 'p2p'}
    assert all('-demc-fx-cpw' in k for k in host), sorted(host)


def test_general_chain_sharded_over_two_ranks(tmp_path):
    kernels = run_ranks_grid('general', 2, tmp_path)
    host = {k for k, v in kernels.items() if v[0] != 'p2p'}
    assert not host, sorted(host)


def test_resident_plans_sharded_over_two_ranks(tmp_path):
    kernels = run_ranks_grid('resident', 2, tmp_path)
    assert all('resident' in kernel for _, kernel in kernels.values()), kernels
    assert all('resident_async' in kernels[k][1] for k in kernels if 'async' in k), kernels
