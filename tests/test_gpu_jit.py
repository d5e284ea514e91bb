"""Run-time specialisation of the one-wave loops (mc3_amd/jit.py, mc3hip_jit_request /
mc3hip_jit_install): polynomials of 5..8 coefficients and design matrices of up to 10 columns
on the hand-laid-out loops, compiled for the very model when the context is created -- against
the oracle, bit for bit, like every other kernel; and against the same run with the
specialisation switched off (the general chain)."""
import os

import numpy as np
import pytest

import helpers as H
from test_gpu_parity import assert_same, run_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from mc3_amd import engine
    return engine


def poly_problem(k, nchains=12, nref=80, ndata=300, **kw):
    """A polynomial of k coefficients around the tutorial's quadratic on x in [0, 1]."""
    prob = H.quad_problem(ndata=ndata, nchains=nchains,
                          nsamples=nchains*nref*kw.get('thinning', 1), **kw)
    pad = k - 3
    prob.update(indep=prob['indep']/10.0,
                params=np.concatenate([[10.0, -20.0, 10.0], np.zeros(pad)]),
                pstep=np.concatenate([[1.0, 5.0, 10.0], np.full(pad, 1.0)]),
                pmin=np.concatenate([[-10.0, -200.0, -1000.0], np.full(pad, -1e4)]),
                pmax=np.concatenate([[40.0, 200.0, 1000.0], np.full(pad, 1e4)]))
    return prob


def linear_problem(k, nchains=12, nref=80, priors=False, ndata=700, **kw):
    prob = H.linear_problem(ndata=ndata, nmodel=k, nchains=nchains,
                            nsamples=nchains*nref*kw.get('thinning', 1), **kw)
    if not priors:
        for key in ('prior', 'priorlow', 'priorup'):
            prob[key] = np.zeros(k)
    return prob


def quad_wide(ndata, nchains=10, nref=80, priors=False, ncoef=3, **kw):
    """The tutorial's quadratic (or a line / a cubic through it) on more than 1024 points."""
    prob = H.quad_problem(ndata=ndata, nchains=nchains,
                          nsamples=nchains*nref*kw.get('thinning', 1), **kw)
    prob.update(params=np.array([10.0, -2.0, 0.1, 0.001])[:ncoef],
                pstep=np.array([1.0, 0.5, 0.1, 0.002])[:ncoef],
                pmin=np.array([-10.0, -20.0, -10.0, -1.0])[:ncoef],
                pmax=np.array([40.0, 20.0, 10.0, 1.0])[:ncoef])
    if priors:
        prob.update(prior=np.array([3.0, 0.0, 0.5, 0.0])[:ncoef],
                    priorlow=np.array([0.5, 0.0, 0.02, 0.0])[:ncoef],
                    priorup=np.array([0.25, 0.0, 0.04, 0.0])[:ncoef])
    else:
        for key in ('prior', 'priorlow', 'priorup'):
            if key in prob:
                prob[key] = np.zeros(ncoef)
    return prob


def with_pstep(prob, pstep):
    pstep = np.array(pstep)
    prob['pstep'] = pstep
    for j in np.where(pstep < 0)[0]:                 # a shared parameter starts at its source's
        prob['params'][j] = prob['params'][int(-pstep[j]) - 1]
    return prob


CASES = {
    'poly5-demc': lambda: poly_problem(5, sampler='demc'),
    'poly8-demc': lambda: poly_problem(8, sampler='demc'),
    'poly6-snooker': lambda: poly_problem(6, sampler='snooker'),
    'poly5-demc-thin': lambda: poly_problem(5, sampler='demc', thinning=3),
    'poly7-demc-lag': lambda: poly_problem(7, sampler='demc', zlag=2),
    'lin3-demc-eps': lambda: linear_problem(3, sampler='demc', fepsilon=0.01),
    'poly6-snooker-thin': lambda: poly_problem(6, sampler='snooker', thinning=2),
    'lin2-demc': lambda: linear_problem(2, sampler='demc'),
    'lin5-demc': lambda: linear_problem(5, sampler='demc'),
    'lin10-demc': lambda: linear_problem(10, sampler='demc'),
    'lin10-snooker': lambda: linear_problem(10, sampler='snooker'),
    'lin3-mrw': lambda: linear_problem(3, sampler='mrw'),
    'lin7-demc-priors': lambda: linear_problem(7, sampler='demc', priors=True),
    'lin10-demc-priors-thin': lambda: linear_problem(10, sampler='demc', priors=True, thinning=2),
    'lin6-snooker-priors': lambda: linear_problem(6, sampler='snooker', priors=True),
    'lin4-demc-eps-priors': lambda: linear_problem(4, sampler='demc', priors=True,
                                                   fepsilon=0.02),
    'lin4-snooker-eps': lambda: linear_problem(4, sampler='snooker', fepsilon=0.02),
    'lin3-snooker-eps-priors': lambda: linear_problem(3, sampler='snooker', priors=True,
                                                      fepsilon=0.01),
    # round 5: the random walk and the support distribution's deviates (fepsilon != 0) for more
    # than four coefficients -- a Philox slot per two of them (chain.py:185,219-220,232)
    'poly5-mrw': lambda: poly_problem(5, sampler='mrw'),
    'poly8-mrw-thin': lambda: poly_problem(8, sampler='mrw', thinning=3),
    'lin6-mrw-priors': lambda: linear_problem(6, sampler='mrw', priors=True),
    'lin10-mrw': lambda: linear_problem(10, sampler='mrw'),
    'lin9-mrw-fixed-shared': lambda: with_pstep(
        linear_problem(9, sampler='mrw'),
        [0.05, 0.0, 0.05, -1.0, 0.05, 0.05, 0.0, 0.05, -3.0]),
    'poly6-demc-eps': lambda: poly_problem(6, sampler='demc', fepsilon=0.01),
    'lin7-demc-eps-priors': lambda: linear_problem(7, sampler='demc', priors=True,
                                                   fepsilon=0.02),
    'lin10-demc-eps': lambda: linear_problem(10, sampler='demc', fepsilon=0.01),
    'poly7-snooker-eps': lambda: poly_problem(7, sampler='snooker', fepsilon=0.01),
    'lin10-snooker-eps-priors': lambda: linear_problem(10, sampler='snooker', priors=True,
                                                       fepsilon=0.02),
    'lin5-snooker-eps-fixed': lambda: with_pstep(
        linear_problem(5, sampler='snooker', fepsilon=0.02), [0.05, 0.05, 0.0, 0.05, 0.05]),
    # round 5: more than 1024 points with what the wide loops of the library do not carry --
    # 2, 4 or 8 tiles in the registers of four waves (lean_*_body<.., NW = 4, NT>)
    'wide-quad3000-demc-fixed': lambda: with_pstep(quad_wide(3000, sampler='demc'),
                                                   [1.0, 0.0, 0.1]),
    'wide-quad2048-demc-eps-priors': lambda: quad_wide(2048, sampler='demc', fepsilon=0.01,
                                                       priors=True),
    'wide-quad5000-snooker-priors': lambda: quad_wide(5000, sampler='snooker', priors=True),
    'wide-quad8192-snooker-eps': lambda: quad_wide(8192, sampler='snooker', fepsilon=0.01),
    'wide-quad7000-mrw': lambda: quad_wide(7000, sampler='mrw'),
    'wide-quad1500-mrw-shared': lambda: with_pstep(quad_wide(1500, sampler='mrw'),
                                                   [1.0, 0.5, -2.0]),
    # ... and 8 193 ... 16 384 points: sixteen tiles in the registers of EIGHT waves
    'wide-quad16384-demc-priors': lambda: quad_wide(16384, sampler='demc', priors=True),
    'wide-quad12000-mrw': lambda: quad_wide(12000, sampler='mrw'),
    'wide-quad9000-snooker-eps-fixed': lambda: with_pstep(
        quad_wide(9000, sampler='snooker', fepsilon=0.01), [1.0, 0.0, 0.1]),
    # as many coefficients as a row of Z has lanes (round 5; before: 8 / 10)
    'poly10-demc': lambda: poly_problem(10, sampler='demc'),
    'poly12-snooker-thin': lambda: poly_problem(12, sampler='snooker', thinning=2),
    'poly16-mrw': lambda: poly_problem(16, sampler='mrw'),
    'lin12-demc-priors': lambda: linear_problem(12, sampler='demc', priors=True),
    'lin16-snooker': lambda: linear_problem(16, sampler='snooker'),
    'lin13-mrw-fixed': lambda: with_pstep(linear_problem(13, sampler='mrw'),
                                          [0.05]*6 + [0.0] + [0.05]*6),
    'lin11-demc-eps': lambda: linear_problem(11, sampler='demc', fepsilon=0.01),
    'wide-poly10-4000-demc': lambda: poly_problem(10, sampler='demc', ndata=4000),
    'wide-lin12-2000-snooker-priors': lambda: linear_problem(12, sampler='snooker', priors=True,
                                                             ndata=2000),
    'wide-lin15-2000-mrw': lambda: linear_problem(15, sampler='mrw', ndata=2000),
    # what was left of the small polynomials: thinning with fixed coefficients / priors, thinning
    # on the wide loops, ensembles too large for the wide loops (one wave, two tiles in registers)
    'left-quad1000-snooker-priors-thin3': lambda: quad_wide(1000, sampler='snooker', priors=True,
                                                            thinning=3),
    'left-quad700-mrw-fixed-thin2': lambda: with_pstep(quad_wide(700, sampler='mrw', thinning=2),
                                                       [1.0, 0.0, 0.1]),
    'left-quad3000-snooker-thin3': lambda: quad_wide(3000, sampler='snooker', thinning=3),
    'left-quad5000-mrw-priors-thin2': lambda: quad_wide(5000, sampler='mrw', priors=True,
                                                        thinning=2),
    'left-quad3000-demc-thin3': lambda: quad_wide(3000, sampler='demc', thinning=3),
    'left-quad8192-demc-priors-thin2': lambda: quad_wide(8192, sampler='demc', priors=True,
                                                         thinning=2),
    'left-poly6-5000-demc-thin4': lambda: poly_problem(6, sampler='demc', ndata=5000, thinning=4),
    'left-quad2048-demc-priors-800chains': lambda: quad_wide(2048, sampler='demc', priors=True,
                                                             nchains=800, nref=80),
    'left-quad3000-snooker-fixed-600chains': lambda: with_pstep(
        quad_wide(3000, sampler='snooker', nchains=600, nref=80), [1.0, 0.0, 0.1]),
    'left-quad4096-mrw-thin2-600chains': lambda: quad_wide(4096, sampler='mrw', thinning=2,
                                                           nchains=600, nref=80),
    # ... and the models that are specialised anyway, beyond one register tile
    'wide-poly6-3000-snooker': lambda: poly_problem(6, sampler='snooker', ndata=3000),
    'wide-poly8-8000-mrw': lambda: poly_problem(8, sampler='mrw', ndata=8000),
    'wide-poly5-12000-demc-eps': lambda: poly_problem(5, sampler='demc', fepsilon=0.01,
                                                      ndata=12000),
    'wide-lin5-2048-demc-priors': lambda: linear_problem(5, sampler='demc', priors=True,
                                                         ndata=2048),
    'wide-lin10-2000-snooker': lambda: linear_problem(10, sampler='snooker', ndata=2000),
    'wide-lin10-4096-mrw': lambda: linear_problem(10, sampler='mrw', ndata=4096),
    'wide-lin4-5000-demc-eps': lambda: linear_problem(4, sampler='demc', fepsilon=0.01,
                                                      ndata=5000),
    'wide-lin2-8000-snooker-priors': lambda: linear_problem(2, sampler='snooker', priors=True,
                                                            ndata=8000),
    'wide-line4096-demc-priors': lambda: quad_wide(4096, sampler='demc', priors=True, ncoef=2),
    'wide-cubic6000-mrw-fixed': lambda: with_pstep(quad_wide(6000, sampler='mrw', ncoef=4),
                                                   [1.0, 0.5, 0.0, 0.002]),
    # fixed and shared coefficients under the helper waves (a conditional gather once handed
    # the helpers zeros for the coefficients behind a fixed one: tools/fuzz_hunt.py)
    'lin8-snooker-fixed': lambda: with_pstep(linear_problem(8, sampler='snooker'),
                                             [0.05, 0.05, 0.05, 0.05, 0.0, 0.05, 0.05, 0.05]),
    'lin6-demc-fixed': lambda: with_pstep(linear_problem(6, sampler='demc'),
                                          [0.05, 0.0, 0.0, 0.0, 0.05, 0.05]),
    'lin10-snooker-fixed-shared-lag': lambda: with_pstep(
        linear_problem(10, sampler='snooker', zlag=1),
        [0.05, 0.0, 0.0, -1.0, -1.0, 0.05, 0.0, 0.05, 0.05, -1.0]),
    'lin9-demc-shared-priors': lambda: with_pstep(
        linear_problem(9, sampler='demc', priors=True),
        [0.05, -1.0, 0.05, 0.0, 0.05, 0.05, 0.0, 0.05, 0.05]),
}


@pytest.mark.parametrize('case', sorted(CASES))
def test_specialised_loops_against_the_oracle(eng, case):
    prob = CASES[case]()
    gpu = eng.DeviceChains(**prob)
    assert gpu.jit_served, 'the library asked for no specialisation'
    orc = H.make_oracle(prob, plan=gpu.plan)
    Z0, lp0 = H.initial_population(orc)
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(nthreads=4)
    for n in (1, 2, 20, 21, 22, 14):            # (several launches: each resumes from Z)
        gpu.run(n)
    plan = gpu.plan
    assert plan['jit'] == 1 and plan['fast'] == 2 and plan['persistent'] > 0, (plan,
                                                                              gpu.jit_served)
    assert plan['recoveries'] == 0
    assert_same(orc, gpu, trace=False)
    gpu.close()


def test_the_request_names_the_loop_and_the_cache_keeps_it(eng, tmp_path, monkeypatch):
    from mc3_amd import jit
    monkeypatch.setenv('MC3HIP_JIT_CACHE', str(tmp_path))
    prob = linear_problem(10, sampler='snooker')
    with eng.DeviceChains(**prob) as gpu:
        # (ten columns do not fit one wave's registers: four waves share the data sum)
        assert gpu.jit_served == ('lean_snooker_body<210, false, false, false, false, false, '
                                  'false, 4> /*threads=256*/')
        assert gpu.plan['jit'] == 1
    made = sorted(os.listdir(str(tmp_path)))
    assert len(made) == 1 and made[0].endswith('.hsaco')
    stamp = os.path.getmtime(os.path.join(str(tmp_path), made[0]))
    with eng.DeviceChains(**prob) as gpu:       # the same shape again: the file, not hipcc
        assert gpu.plan['jit'] == 1
    assert os.path.getmtime(os.path.join(str(tmp_path), made[0])) == stamp
    # shapes the library serves itself, or no loop serves, ask for nothing
    with eng.DeviceChains(**H.quad_problem(sampler='demc')) as gpu:
        assert gpu.jit_request() == '' and gpu.plan['jit'] == 0 and gpu.plan['fast'] == 2
    with eng.DeviceChains(**H.wavelet_problem()) as gpu:
        assert gpu.jit_request() == ''
    with eng.DeviceChains(**linear_problem(17, sampler='demc')) as gpu:     # too many columns
        assert gpu.jit_request() == '' and gpu.plan['fast'] == 0
    with eng.DeviceChains(**poly_problem(5, sampler='mrw')) as gpu:  # (round 5: served too)
        assert gpu.jit_served == 'lean_mrw_body<5, false, false, false, false, false>'
    assert jit.KERNEL_NAME == 'mc3_jit_kernel'


@pytest.mark.parametrize('case', ['poly5-demc', 'lin10-demc'])
def test_without_the_specialisation_the_general_chain_gives_the_same_rows(eng, monkeypatch, case):
    prob = CASES[case]()
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['jit'] == 1
    Z, lp, _ = gpu.rows()
    gpu.close()
    monkeypatch.setenv('MC3HIP_NO_JIT', '1')
    orc2, gpu2 = run_pair(eng, prob, trace=False)
    assert gpu2.plan['jit'] == 0 and gpu2.plan['fast'] == 0 and not gpu2.jit_served
    Z2, lp2, _ = gpu2.rows()
    gpu2.close()
    np.testing.assert_array_equal(Z, Z2)
    np.testing.assert_array_equal(lp, lp2)
    np.testing.assert_array_equal(Z, orc.Z)


def test_a_specialised_launch_that_gives_up_is_redone(eng, monkeypatch):
    monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '21')
    prob = CASES['lin5-demc']()
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['recoveries'] == 1
    assert_same(orc, gpu, trace=False)
    gpu.close()


def test_more_chains_than_the_specialised_loop_holds_take_turns_on_the_general_chain(
        eng, monkeypatch):
    monkeypatch.setenv('MC3HIP_PERSIST_CAPACITY', '5')
    prob = linear_problem(5, nchains=16, nref=30, sampler='demc')
    orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert plan['persistent'] > 0 and plan['chains_per_workgroup'] > 1 and plan['jit'] == 0, plan
    assert_same(orc, gpu, trace=False)
    gpu.close()


def test_a_plan_that_depends_on_the_specialisation_needs_the_callers_promise(eng, monkeypatch):
    """16 384 points, a quadratic with Gaussian priors, 64 chains: with the promise to specialise (MC3HIP_FLAG_JIT,
    what DeviceChains passes when a compiler is at hand) the library plans tiles in the
    registers of eight waves per chain; a caller that does not promise (flags = 0: jit=False) --
    or whose compiler fails -- gets the plan the library serves with its own instances (the
    one-launch tiled plan), not the general chain behind a layout made for a loop that never
    came.  All three runs give the oracle's rows."""
    from mc3_amd import jit
    prob = quad_wide(16384, nchains=64, nref=12, priors=True)
    orc, gpu = run_pair(eng, prob, trace=False)
    served = gpu.plan
    assert served['jit'] == 1 and served['tile'] == 0, served
    assert_same(orc, gpu, trace=False)
    gpu.close()
    with eng.DeviceChains(jit=False, **prob) as gpu:
        plain = gpu.plan
        assert plain['jit'] == 0 and plain['tile'] == 4096, plain
        orc2 = H.make_oracle(prob, plan=plain)
        Z0, lp0 = H.initial_population(orc2)
        orc2.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc2.run(nthreads=4)
        gpu.run()
        assert gpu.plan['resident'] == 1, gpu.plan
        assert_same(orc2, gpu, trace=False)

    def broken(request, verbose=False):
        raise RuntimeError('no compiler today')
    monkeypatch.setattr(jit, 'code_object', broken)
    with eng.DeviceChains(**prob) as gpu:
        assert not gpu.jit_served
        assert gpu.plan['jit'] == 0 and gpu.plan['tile'] == 4096, gpu.plan
        gpu.set_population(Z0, lp0)
        gpu.run()
        assert_same(orc2, gpu, trace=False)
