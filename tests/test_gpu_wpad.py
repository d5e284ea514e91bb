"""Wavelet likelihood of a series whose length is no power of two (mc3hip_problem.wl_nreal): the
caller pads data / uncert / indep to 2^M points and says how many are data; the residuals of
the padding are exactly zero -- the zero padding of src_c/_dwt.c:82-91 made well defined (the
reference then transforms the unpadded length, which only 2^M survives).  Every wavelet geometry
against the lockstep oracle, bit for bit, and the chi-squared itself against the
reference-faithful kernel (oracle/oracle_kernels.c) fed residuals that vanish on the padding."""
import numpy as np
import pytest

import helpers as H
from test_gpu_parity import assert_same, run_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from mc3_amd import engine
    return engine


def padded_problem(nreal, nchains=8, nref=30, sampler='snooker', model='poly', **kw):
    """helpers.wavelet_problem at `nreal` points, padded to the next power of two: data 0,
    uncert 1, the abscissae repeat the last one."""
    n2 = 1 << int(np.ceil(np.log2(nreal)))
    prob = H.wavelet_problem(ndata=nreal, nchains=nchains, nsamples=nchains*nref, sampler=sampler,
                             **kw)
    pad = n2 - nreal
    prob['data'] = np.concatenate([prob['data'], np.zeros(pad)])
    prob['uncert'] = np.concatenate([prob['uncert'], np.ones(pad)])
    prob['indep'] = np.concatenate([prob['indep'], np.full(pad, prob['indep'][-1])])
    if model == 'linear':                      # the same quadratic as a design matrix
        x = prob['indep']
        prob.update(indep=np.array([np.ones_like(x), x, x*x]), model='linear', nmodel=3)
    prob['wl_nreal'] = nreal
    return prob, n2


@pytest.mark.parametrize('nreal,sampler,model', [
    (50, 'snooker', 'poly'), (1000, 'demc', 'poly'), (1000, 'snooker', 'linear'),
    (3000, 'snooker', 'poly'), (4095, 'mrw', 'poly'), (6000, 'snooker', 'poly'),
    (10000, 'demc', 'poly'), (12345, 'snooker', 'linear'), (16383, 'snooker', 'poly'),
    (20000, 'snooker', 'poly'), (3, 'snooker', 'poly'), (8193, 'demc', 'poly')])
def test_padded_series_against_the_oracle(eng, nreal, sampler, model):
    prob, n2 = padded_problem(nreal, sampler=sampler, model=model,
                              nref=12 if nreal > 8192 else 30)
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['persistent'] > 0 or n2 <= 4
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('nreal', [50, 1000, 6000, 12345, 20000])
def test_padded_chisq_is_the_reference_kernel_on_zero_padded_residuals(eng, nreal):
    from oracle import kernels as K
    prob, n2 = padded_problem(nreal)
    with eng.DeviceChains(**prob) as gpu:
        rs = np.random.RandomState(nreal)
        for _ in range(4):
            p = prob['params'] + rs.normal(0, 0.05, len(prob['params']))*(prob['pstep'] > 0)
            got = gpu.eval_chisq(p[None, :])[0]
            model = H.quad(p[:3], prob['indep'])
            model[nreal:] = prob['data'][nreal:]           # residuals vanish on the padding
            want = K.dwt_chisq(model, prob['data'], p)
            np.testing.assert_allclose(got, want, rtol=1e-10)
        # ... and the padding matters: the unpadded flag gives another number
    prob2 = dict(prob, wl_nreal=0)
    with eng.DeviceChains(**prob2) as gpu:
        other = gpu.eval_chisq(prob['params'][None, :])[0]
    assert abs(other - want) > 1e-6*abs(want)


def test_bad_counts_are_refused(eng):
    prob, n2 = padded_problem(50)
    with pytest.raises(Exception):
        eng.DeviceChains(**dict(prob, wl_nreal=n2 + 1))
    with pytest.raises(Exception):
        eng.DeviceChains(**dict(prob, wl_nreal=-1))


@pytest.mark.parametrize('ndata,sampler', [(1000, 'snooker'), (5000, 'demc')])
def test_sample_with_wpad_equals_the_cpu_checker(ndata, sampler):
    """mc3_amd.sample(..., wlike=True, wpad=True) end to end on the GPU == the same call with the
    CPU checker injected; without the flag the series is refused."""
    import mc3_amd
    from test_gpu_api import assert_same_dict, both, quad
    prob = H.wavelet_problem(ndata=ndata)
    kw = dict(data=prob['data'], uncert=prob['uncert'], func=quad, params=prob['params'],
              indparams=[prob['indep']], pstep=prob['pstep'], pmin=prob['pmin'],
              pmax=prob['pmax'], sampler=sampler, nsamples=700, burnin=10, nchains=7, wlike=True)
    with pytest.raises(ValueError, match='power-of-two'):
        both(**kw)
    gpu, cpu = both(wpad=True, **kw)
    assert_same_dict(gpu, cpu)
    assert gpu['best_model'].shape == (ndata,)


@pytest.mark.parametrize('seed', range(60))
def test_random_wavelet_problem_with_a_padded_tail(eng, seed):
    """test_gpu_fuzz.random_problem's wavelet problems (either model kind, 2 .. 2^14 points, every
    sampler, thinning, fixed / shared parameters) with only the first wl_nreal points data."""
    import test_gpu_fuzz as F
    prob = F.random_problem(31000 + seed, wavelet_any_model=True)
    rs = np.random.RandomState(31000 + seed)
    n = len(prob['data'])
    prob['wl_nreal'] = int(rs.randint(n//2 + 1, n)) if n > 2 else 1
    gpu = eng.DeviceChains(**prob)
    orc = H.make_oracle(prob, plan=gpu.plan)
    try:
        Z0, lp0 = H.initial_population(orc, seed=seed)
    except ValueError:
        gpu.close()
        pytest.skip('no finite initial population')
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(nthreads=4)
    gpu.run(max(1, gpu.nzchain//3))
    gpu.run()
    F.compare_final(gpu, orc, f'seed {seed} nreal {prob["wl_nreal"]} of {n}')
    gpu.close()
