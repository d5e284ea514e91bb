"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle.

Bar (north_star): accept counts bit-exact under a fixed seed, chisq within 1e-10
relative of the reference.  The lockstep oracle is built to the same arithmetic
contract as the kernels, so here everything -- Z rows, log-posteriors, chain
ids, accept / out-of-bounds counters, per-step chisq and flags -- is compared
for exact equality; the reference's own numbers (golden vectors generated from
its compiled C) are compared at 1e-10.
"""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from mc3_amd import engine
    return engine


def run_pair(eng, prob, trace=True, nrefresh=None, **gpu_kw):
    thin = prob.get('thinning', 1)
    niter = int(np.ceil(prob['nsamples']/prob['nchains']/thin))*thin
    gpu = eng.DeviceChains(trace_iters=niter if trace else 0, **prob, **gpu_kw)
    # the oracle sums the data in the order ("plan") the library reports
    orc = H.make_oracle(prob, plan=gpu.plan)
    Z0, lp0 = H.initial_population(orc)
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    assert gpu.plan['vlanes'] == orc.P.vlanes
    orc.run(nrefresh=nrefresh, trace=trace, nthreads=4)
    gpu.run(nrefresh=nrefresh)
    return orc, gpu


def assert_same(orc, gpu, trace=True):
    Z, lp, zc = gpu.rows()
    st = gpu.state()
    if trace:
        tc, tf = gpu.trace(orc.trace_flag.shape[0])
        np.testing.assert_array_equal(tf, orc.trace_flag)
        np.testing.assert_array_equal(tc, orc.trace_chisq)
    np.testing.assert_array_equal(zc, orc.zchain)
    np.testing.assert_array_equal(Z, orc.Z)
    np.testing.assert_array_equal(lp, orc.log_post)
    np.testing.assert_array_equal(st['numaccept'], orc.numaccept)
    np.testing.assert_array_equal(st['outbounds'], orc.outbounds)
    np.testing.assert_array_equal(st['X'], orc.X)
    np.testing.assert_array_equal(st['chisq'], orc.chisq)
    np.testing.assert_array_equal(st['best_chisq'], orc.best_chisq)
    np.testing.assert_array_equal(st['best_iter'], orc.best_iter)
    acc = st['best_iter'] >= 0
    np.testing.assert_array_equal(st['bestx'][acc], orc.bestx[acc])
    assert st['numaccept'].sum() > 0


# ---- building blocks -------------------------------------------------------
def test_device_math_matches_host_bit_for_bit(eng):
    import ctypes
    from mc3_amd import _lib
    from oracle import kernels as ok
    L, O = _lib.load(), ok.lib()
    rs = np.random.RandomState(0)
    cases = {
        0: (np.concatenate([rs.uniform(-745, 710, 50000), rs.normal(0, 1, 50000)]), O.ls_det_exp),
        1: (np.exp(rs.uniform(-700, 700, 100000)), O.ls_det_log),
        2: (rs.uniform(-60, 60, 100000), O.ls_det_exp2),
    }
    for op, (xs, host) in cases.items():
        xs = np.ascontiguousarray(xs)
        out = np.empty_like(xs)
        _lib.check(L.mc3hip_selftest_math(0, op, _lib.dptr(xs), len(xs), _lib.dptr(out)))
        want = np.array([host(float(x)) for x in xs])
        np.testing.assert_array_equal(out, want)
    us = np.ascontiguousarray(rs.uniform(0, 1, 100000))
    s, c = ctypes.c_double(), ctypes.c_double()
    ws, wc = np.empty_like(us), np.empty_like(us)
    for i, u in enumerate(us):
        O.ls_det_sincos2pi(float(u), ctypes.byref(s), ctypes.byref(c))
        ws[i], wc[i] = s.value, c.value
    for op, want in ((3, ws), (4, wc)):
        out = np.empty_like(us)
        _lib.check(L.mc3hip_selftest_math(0, op, _lib.dptr(us), len(us), _lib.dptr(out)))
        np.testing.assert_array_equal(out, want)
    # IEEE sqrt and division are correctly rounded on the device
    xs = np.ascontiguousarray(np.exp(rs.uniform(-300, 300, 100000)))
    out = np.empty_like(xs)
    _lib.check(L.mc3hip_selftest_math(0, 5, _lib.dptr(xs), len(xs), _lib.dptr(out)))
    np.testing.assert_array_equal(out, np.sqrt(xs))
    _lib.check(L.mc3hip_selftest_math(0, 6, _lib.dptr(xs), len(xs), _lib.dptr(out)))
    np.testing.assert_array_equal(out, 1.0/xs)


def test_device_philox_known_answers(eng):
    from mc3_amd import _lib
    from oracle import lockstep as ls
    L = _lib.load()
    rs = np.random.RandomState(1)
    ctr = rs.randint(0, 2**32, (1000, 4), dtype=np.uint64).astype(np.uint32)
    ctr[0] = 0
    ctr[1] = 0xffffffff
    ctr[2] = [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]
    up = _lib._up
    for key in ((0, 0), (0xffffffff, 0xffffffff), (0xa4093822, 0x299f31d0)):
        k = np.array(key, np.uint32)
        out = np.empty_like(ctr)
        _lib.check(L.mc3hip_selftest_philox(0, ctr.ctypes.data_as(up), len(ctr),
                                            k.ctypes.data_as(up), out.ctypes.data_as(up)))
        for i in range(0, len(ctr), 37):
            assert tuple(out[i]) == ls.philox4x32_10(tuple(int(v) for v in ctr[i]), key)
        if key == (0, 0):
            assert tuple(out[0]) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
        if key == (0xffffffff, 0xffffffff):
            assert tuple(out[1]) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
        if key == (0xa4093822, 0x299f31d0):
            assert tuple(out[2]) == (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_draws_are_rocrand_philox_state_word_for_word(eng):
    """csrc/philox.h:7-9 claims that a draw for (iteration, chain, slot) under `seed` is what
    rocRAND's rocrand_state_philox4x32_10 yields after rocrand_init(seed, chain | slot << 32,
    4*iteration): draw from rocRAND ITSELF (tools/probes/rocrand_philox_probe.hip, a child
    process that includes rocrand_kernel.h) and compare with the library's rounds and the
    checker's, word for word."""
    import subprocess
    from mc3_amd import _lib, build
    from oracle import lockstep as ls
    exe = build.probe_path('rocrand_philox_probe')
    assert os.path.exists(exe), 'tools/probes/_bin/rocrand_philox_probe: run __graft_entry__.build()'
    L = _lib.load()
    rs = np.random.RandomState(11)
    n = 4096
    req = np.empty((n, 3), np.uint64)
    req[:, 0] = rs.randint(0, 2**40, n, dtype=np.uint64)        # iteration (beyond 32 bits too)
    req[:, 1] = rs.randint(0, 2**20, n, dtype=np.uint64)        # global chain id
    req[:, 2] = rs.randint(0, 8, n, dtype=np.uint64)            # slot
    req[0] = (0, 0, 0)
    req[1] = (2**32 - 1, 2**20 - 1, 7)
    req[2] = (2**32, 1, 0)                                       # carry into the counter's word 1
    req[3] = (2**62 - 1, 255, 4)                                 # (4*iteration must not overflow)
    up = _lib._up
    for seed in (0, 12345, 0xdeadbeefcafef00d):
        blob = np.array([seed, n], np.uint64).tobytes() + req.tobytes()
        res = subprocess.run([exe], input=blob, stdout=subprocess.PIPE, timeout=120, check=True)
        rocrand = np.frombuffer(res.stdout, np.uint32).reshape(n, 4)
        # the library's counter layout: (iteration lo, iteration hi, chain, slot), key = seed
        ctr = np.empty((n, 4), np.uint32)
        ctr[:, 0] = (req[:, 0] & np.uint64(0xffffffff)).astype(np.uint32)
        ctr[:, 1] = (req[:, 0] >> np.uint64(32)).astype(np.uint32)
        ctr[:, 2] = req[:, 1].astype(np.uint32)
        ctr[:, 3] = req[:, 2].astype(np.uint32)
        key = np.array([seed & 0xffffffff, seed >> 32], np.uint32)
        out = np.empty_like(ctr)
        _lib.check(L.mc3hip_selftest_philox(0, ctr.ctypes.data_as(up), n,
                                            key.ctypes.data_as(up), out.ctypes.data_as(up)))
        np.testing.assert_array_equal(out, rocrand)
        for i in range(0, n, 97):
            assert tuple(rocrand[i]) == ls.philox4x32_10(tuple(int(v) for v in ctr[i]),
                                                         (int(key[0]), int(key[1])))


# ---- likelihood kernels against the reference's golden vectors --------------
@pytest.mark.parametrize('ndata', [8, 32, 1000, 16384])
def test_gauss_chisq_vs_reference_vectors(eng, golden_dir, ndata):
    kv = np.load(os.path.join(golden_dir, 'kernel_vectors.npz'))
    x, data, uncert = H.tutorial_data(ndata, 1000 + ndata)
    P = kv[f'gauss_{ndata}_params']
    base = dict(data=data, uncert=uncert, indep=x, params=P[0], pstep=np.ones(3),
                nchains=4, nsamples=40, sampler='demc')
    with eng.DeviceChains(**base) as g:
        got = g.eval_chisq(P)
        np.testing.assert_allclose(got, kv[f'gauss_{ndata}_chisq'], rtol=1e-10)
        model = g.eval_model_values(P[0])
        np.testing.assert_allclose(model, H.quad(P[0], x), rtol=1e-13, atol=1e-13)
    pri = dict(prior=kv['gauss_prior'], priorlow=kv['gauss_priorlow'],
               priorup=kv['gauss_priorup'])
    with eng.DeviceChains(**base, **pri) as g:
        got = g.eval_chisq(P)
        np.testing.assert_allclose(got, kv[f'gauss_{ndata}_chisq_priors'], rtol=1e-10)
        orc = H.make_oracle(dict(base, **pri))
        np.testing.assert_array_equal(got, orc.eval_chisq(P))


@pytest.mark.parametrize('ndata', [2, 4, 8, 32, 1024, 16384])
def test_wavelet_chisq_vs_reference_vectors(eng, golden_dir, ndata):
    kv = np.load(os.path.join(golden_dir, 'kernel_vectors.npz'))
    rs2 = np.random.RandomState(2000 + ndata)
    x = np.linspace(0, 10, ndata)
    data = H.quad([3.0, -2.4, 0.5], x) + rs2.normal(0, 1.0, ndata) \
        + 0.5*np.sin(2*np.pi*x)
    P = kv[f'dwt_{ndata}_params']
    base = dict(data=data, uncert=np.ones(ndata), indep=x, params=P[0],
                pstep=np.array([1.0, 1, 1, 0, 1, 1]), nchains=4, nsamples=40,
                sampler='snooker', wlike=True)
    # the geometry of the blocked pyramid is a function of ndata alone (1, 4 or 8 waves
    # per chain); the checker follows the one the plan reports
    with eng.DeviceChains(**base) as g:
        plan = g.plan
        assert plan['waves_per_chain'] == (8 if ndata > 4096 else 4 if ndata > 2048 else 1)
        got = g.eval_chisq(P)
    np.testing.assert_allclose(got, kv[f'dwt_{ndata}_chisq'], rtol=1e-10)
    np.testing.assert_array_equal(got, H.make_oracle(base, plan=plan).eval_chisq(P))
    pri = dict(prior=kv['dwt_prior'], priorlow=kv['dwt_priorlow'],
               priorup=kv['dwt_priorup'])
    with eng.DeviceChains(**base, **pri) as g:
        np.testing.assert_allclose(g.eval_chisq(P), kv[f'dwt_{ndata}_chisq_priors'],
                                   rtol=1e-10)


def test_known_answers_of_the_reference_tests(eng):
    """tests/test_stats.py:135-171 of the reference, through the device."""
    data = np.array([1.1, 1.2, 0.9, 1.0])
    uncert = np.full(4, 0.1)
    base = dict(data=data, uncert=uncert, indep=np.zeros(4), nchains=4, nsamples=40,
                sampler='demc')
    with eng.DeviceChains(params=np.array([1.0]), pstep=np.array([1.0]), **base) as g:
        np.testing.assert_allclose(g.eval_chisq(np.array([1.0])), 6.0, rtol=1e-13)
    # a constant model (nmodel=1) with two extra parameters carrying the priors
    with eng.DeviceChains(params=np.array([1.0, 2.5, 5.5]), pstep=np.ones(3), nmodel=1,
                          prior=np.array([0.0, 2.0, 5.0]),
                          priorlow=np.array([0.0, 0.0, 1.0]),
                          priorup=np.array([0.0, 0.0, 1.0]), **base) as g:
        np.testing.assert_allclose(g.eval_chisq(np.array([1.0, 2.5, 5.5])), 6.25,
                                   rtol=1e-13)
    data = np.array([2.0, 0.0, 3.0, -2.0, -1.0, 2.0, 2.0, 0.0])
    wb = dict(data=data, uncert=np.ones(8), indep=np.zeros(8), nchains=4, nsamples=40,
              sampler='snooker', wlike=True, pstep=np.array([1.0, 0, 1, 1]))
    p = np.array([1.0, 1.0, 0.1, 0.1])
    with eng.DeviceChains(params=p, **wb) as g:
        np.testing.assert_allclose(g.eval_chisq(p), 1693.22308882, rtol=1e-10)
    with eng.DeviceChains(params=p, prior=np.array([0, 1.0, 0.2, 0.3]),
                          priorlow=np.array([0, 0.0, 0.0, 0.1]),
                          priorup=np.array([0, 0.0, 0.0, 0.1]), **wb) as g:
        np.testing.assert_allclose(g.eval_chisq(p), 1697.2230888243134, rtol=1e-10)


@pytest.mark.parametrize('n', [4, 8, 32, 100, 1024])
def test_daub4_vs_reference_vectors(eng, golden_dir, n):
    kv = np.load(os.path.join(golden_dir, 'kernel_vectors.npz'))
    v = kv[f'daub4_{n}_in']
    np.testing.assert_allclose(eng.dwt_daub4(v), kv[f'daub4_{n}_fwd'],
                               rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(eng.dwt_daub4(v, True), kv[f'daub4_{n}_inv'],
                               rtol=1e-12, atol=1e-13)
    # round trip (reference tests/test_stats.py:296-302)
    nw = len(kv[f'daub4_{n}_fwd'])
    pad = np.zeros(nw)
    pad[:n] = v
    np.testing.assert_allclose(eng.dwt_daub4(eng.dwt_daub4(v, True)), pad, atol=1e-10)


# ---- the chain update: bit-exact against the lockstep oracle ---------------
@pytest.mark.parametrize('sampler,extra', [
    ('demc', {}),
    ('snooker', {}),
    ('mrw', {}),
    ('demc', {'fepsilon': 0.01, 'fgamma': 0.9, 'thinning': 2}),
    ('snooker', {'fepsilon': 0.02, 'thinning': 3}),
])
def test_quad_chains_bit_exact(eng, sampler, extra):
    prob = H.quad_problem(sampler=sampler, nchains=9, nsamples=9*150*extra.get('thinning', 1),
                          **extra)
    orc, gpu = run_pair(eng, prob)
    assert_same(orc, gpu)
    gpu.close()


@pytest.mark.parametrize('sampler', ['snooker', 'demc'])
def test_shared_fixed_priors_bounds_bit_exact(eng, sampler):
    prob = H.shared_problem(sampler=sampler)
    orc, gpu = run_pair(eng, prob)
    assert_same(orc, gpu)
    assert gpu.state()['outbounds'].sum() > 0      # the bounds do bite
    gpu.close()


@pytest.mark.parametrize('ndata', [2, 4, 8, 16, 64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_wavelet_chains_bit_exact(eng, ndata):
    prob = H.wavelet_problem(ndata=ndata, nsamples=7*60)
    orc, gpu = run_pair(eng, prob)
    assert_same(orc, gpu)
    gpu.close()


@pytest.mark.parametrize('ndata', [2, 4, 8, 16, 64, 100, 128, 256, 512, 1024, 2048, 4096])
@pytest.mark.parametrize('case', ['plain', 'tight', 'wide', 'resume', 'abort', 'switch'])
def test_wavelet_one_wave_chain_by_hand(eng, monkeypatch, ndata, case):
    """wavelet_one_kernel (round 5): up to 2 048 points, one wave per chain (4 096: four), snooker -- the chain
    written out by hand around wavelet_body<1> (state in registers, draws 21 turns at a time,
    rows a turn ahead) where the general chain ran: identical rows, counters, best fit to the
    checker for every block size, with bounds that reject a third of the proposals, with none,
    across launches (160 refreshes: the batch of draws is refilled seven times), through a
    give-up, and behind the switch the general chain (the same results).  A series whose length
    is no power of two (100 points) is padded by the caller (wl_nreal)."""
    if ndata == 100:
        prob = H.wavelet_problem(ndata=128, nchains=6, nsamples=6*160)
        prob.update(wl_nreal=100)
    else:
        prob = H.wavelet_problem(ndata=ndata, nchains=6, nsamples=6*160)
    if case == 'tight':
        prob.update(pmin=np.array([2.6, -2.6, 0.47, 0.0, 0.3, 0.8]),
                    pmax=np.array([3.4, -2.2, 0.53, 2.0, 0.7, 1.2]))
    if case == 'wide':
        prob.update(pmin=np.full(6, -1e3), pmax=np.full(6, 1e3))
    if case == 'switch':
        monkeypatch.setenv('MC3HIP_NO_WL_ONE', '1')
    if case == 'abort':
        monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '40')
    if case == 'resume':
        monkeypatch.setenv('MC3HIP_PERSIST_REFS', '23')
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 50, 107):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    if case == 'abort':
        assert plan['recoveries'] == 1, plan
    else:
        # (4 096 points: four waves per chain, the chain wave 0's)
        assert plan['persistent'] > 0 and plan['recoveries'] == 0 and \
            plan['waves_per_chain'] == (4 if ndata == 4096 else 1), plan
        assert plan['fast'] == (0 if case == 'switch' else 3), plan
    assert_same(orc, gpu, trace=False)
    if case == 'tight' and ndata >= 64:
        assert gpu.state()['outbounds'].sum() > 50
    gpu.close()


@pytest.mark.parametrize('sampler', ['mrw', 'demc', 'mrw-eps', 'demc-eps', 'snooker-eps'])
@pytest.mark.parametrize('ndata', [2, 16, 100, 256, 1024, 2048, 4096])
@pytest.mark.parametrize('case', ['plain', 'tight', 'resume', 'abort', 'switch'])
def test_wavelet_one_wave_chain_other_samplers(eng, monkeypatch, ndata, sampler, case):
    """wavelet_one_other_kernel: the same one-wave chain for the samplers that need no history --
    the random walk (with a support-distribution factor too: it plays no part in chain.py:219-220)
    and DEMC (the partners' rows of the refresh before, handed over in the launch) -- and the
    support distribution (chain.py:185,217) in DEMC's and snooker's one-wave chains: identical
    rows, counters, best fit to the checker, with bounds that bite, across launches, through a
    give-up, and behind the switch the general chain."""
    smp = sampler.split('-')[0]
    if ndata == 100:
        prob = H.wavelet_problem(ndata=128, nchains=6, nsamples=6*120, sampler=smp)
        prob.update(wl_nreal=100)
    else:
        prob = H.wavelet_problem(ndata=ndata, nchains=6, nsamples=6*120, sampler=smp)
    if sampler.endswith('-eps'):
        prob.update(fepsilon=0.01)
    if case == 'tight':
        prob.update(pmin=np.array([2.6, -2.6, 0.47, 0.0, 0.3, 0.8]),
                    pmax=np.array([3.4, -2.2, 0.53, 2.0, 0.7, 1.2]))
    if case == 'switch':
        monkeypatch.setenv('MC3HIP_NO_WL_ONE', '1')
    if case == 'abort':
        monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '40')
    if case == 'resume':
        monkeypatch.setenv('MC3HIP_PERSIST_REFS', '23')
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 50, 67):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    if case == 'abort':
        assert plan['recoveries'] == 1, plan
    else:
        assert plan['persistent'] > 0 and plan['recoveries'] == 0 and \
            plan['waves_per_chain'] == (4 if ndata == 4096 else 1), plan
        assert plan['fast'] == (0 if case == 'switch' else 3), plan
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('ndata', [32768, 131072])
def test_wavelet_beyond_2_14_points(eng, ndata):
    """More than 32 points per thread: strided pre-levels through global memory bring the
    pyramid down to 2^14 values, then the blocked pyramid runs (src_c/_dwt.c:82-94 takes any
    length).  chisq against the reference-faithful kernels at 1e-10, chains bit-exact
    against the oracle, persistent and per-refresh launches."""
    from oracle import kernels as ok
    prob = H.wavelet_problem(ndata=ndata, nchains=5, nsamples=5*12, sampler='snooker')
    orc, gpu = run_pair(eng, prob)
    assert gpu.plan['waves_per_chain'] == 8 and gpu.plan['persistent'] > 0
    assert_same(orc, gpu)
    rs = np.random.RandomState(3)
    P = prob['params'] + rs.normal(0, 1, (300, 6))*prob['pstep']
    got = gpu.eval_chisq(P)                        # (more vectors than scratch blocks)
    np.testing.assert_array_equal(got, orc.eval_chisq(P))
    want = [ok.dwt_chisq(H.quad(p[:3], prob['indep']), prob['data'], p) for p in P[:20]]
    np.testing.assert_allclose(got[:20], want, rtol=1e-10)
    gpu.close()
    orc2, gpu2 = run_pair(eng, prob, no_persist=True)
    assert_same(orc2, gpu2)
    gpu2.close()


@pytest.mark.parametrize('sampler', ['demc', 'mrw'])
@pytest.mark.parametrize('ndata', [32, 512, 4096, 16384])
def test_wavelet_other_samplers_bit_exact(eng, ndata, sampler):
    prob = H.wavelet_problem(ndata=ndata, nsamples=7*30, sampler=sampler)
    orc, gpu = run_pair(eng, prob)
    assert_same(orc, gpu)
    gpu.close()


def test_linear_model_chains_bit_exact(eng):
    prob = H.linear_problem()
    orc, gpu = run_pair(eng, prob)
    assert_same(orc, gpu)
    gpu.close()


@pytest.mark.parametrize('wpc', ['1', '4'])
def test_waves_per_chain_do_not_change_results(eng, wpc):
    prob = H.quad_problem(ndata=3000, nchains=8, nsamples=8*40, sampler='snooker')
    os.environ['MC3HIP_WPC'] = wpc
    try:
        orc, gpu = run_pair(eng, prob)
    finally:
        del os.environ['MC3HIP_WPC']
    assert gpu.plan['waves_per_chain'] == int(wpc)
    assert_same(orc, gpu)
    gpu.close()


def test_launch_modes_agree_and_resume(eng):
    """600 refreshes through the three launch structures -- eager (one launch per
    refresh), hipGraph (two 256-node replays + an eager tail) and the persistent
    kernel (in-launch hand-off of the Z rows) -- split into several run() calls;
    all equal to the oracle."""
    prob = H.quad_problem(nchains=8, nsamples=8*600, sampler='demc')
    orc, eager = run_pair(eng, prob, trace=False, no_graph=True, no_persist=True)
    assert eager.plan['graph_nodes'] == 0 and eager.plan['persistent'] == 0
    Z0, lp0 = H.initial_population(orc)
    graph = eng.DeviceChains(no_persist=True, **prob)
    assert graph.plan['graph_nodes'] == 256 and graph.plan['persistent'] == 0
    persist = eng.DeviceChains(**prob)
    assert persist.plan['persistent'] > 0
    for g in (graph, persist):
        g.set_population(Z0, lp0)
        g.run(300)
        g.run(37)
        g.run(263)
    for g in (eager, graph, persist):
        assert_same(orc, g, trace=False)
    with pytest.raises(Exception, match='cannot run'):
        graph.run(1)
    for g in (eager, graph, persist):
        g.close()


@pytest.mark.parametrize('builder,kw', [
    (H.quad_problem, dict(sampler='snooker', nchains=9, nsamples=9*150)),
    (H.quad_problem, dict(sampler='demc', nchains=9, nsamples=9*300, thinning=2,
                          fepsilon=0.01)),
    (H.shared_problem, dict()),
    (H.wavelet_problem, dict(nsamples=7*60)),
    (H.linear_problem, dict()),
])
def test_per_refresh_launch_path_bit_exact(eng, builder, kw):
    """The one-launch-per-refresh path (what multi-GPU runs use) on every model /
    sampler family; the default tests above exercise the persistent kernel."""
    prob = builder(**kw)
    orc, gpu = run_pair(eng, prob, no_persist=True)
    assert gpu.plan['persistent'] == 0
    assert_same(orc, gpu)
    gpu.close()


def test_persistent_kernel_many_chains(eng):
    """2048 chains: eight single-wave workgroups per CU, all resident (the full
    occupancy the runtime reports); were they not, the launch would give up and the
    library would redo the run with per-refresh launches."""
    prob = H.quad_problem(ndata=256, nchains=2048, nsamples=2048*60, sampler='demc')
    orc, gpu = run_pair(eng, prob, trace=False)
    assert_same(orc, gpu, trace=False)
    assert gpu.plan['persistent'] > 0
    gpu.close()


@pytest.mark.parametrize('sampler', ['demc', 'snooker'])
def test_config_c3_ensemble_in_one_context(eng, sampler):
    """BASELINE config C3's ensemble -- 2048 chains x ndata 1000 -- in ONE context (what
    eight ranks must reproduce, tests/test_gpu_p2p.py): lean kernels at full occupancy,
    bit-exact against the oracle."""
    prob = H.quad_problem(ndata=1000, nchains=2048, nsamples=2048*40, sampler=sampler)
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['persistent'] > 0 and gpu.plan['fast'] == 2      # (both hand-laid-out loops)
    assert_same(orc, gpu, trace=False)
    assert gpu.plan['recoveries'] == 0
    gpu.close()


@pytest.mark.parametrize('ndata,nchains', [(2000, 1100), (4096, 1030)])
def test_multi_tile_lean_kernels_at_their_own_occupancy(eng, ndata, nchains):
    """The MULTI instances of the lean kernels hold two data tiles in registers: four
    single-wave workgroups per CU, not eight.  More than 1024 chains of such a problem are
    co-resident only because the grid is sized by the occupancy of the instance that is
    launched (not of its plain sibling): the run must complete on the persistent path, without a
    give-up and redo."""
    prob = H.quad_problem(ndata=ndata, nchains=nchains, nsamples=nchains*12, sampler='demc')
    orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert_same(orc, gpu, trace=False)
    assert gpu.plan['recoveries'] == 0, (plan, gpu.plan)
    gpu.close()


@pytest.mark.parametrize('sampler', ['demc', 'snooker', 'mrw'])
def test_lean_instances_with_a_fixed_coefficient(eng, sampler):
    """A fixed polynomial coefficient (pstep = 0) keeps the problem on the lean persistent
    instances (FX instantiations: coefficient j is a free parameter or a fixed value)."""
    prob = H.quad_problem(ndata=500, nchains=64, nsamples=64*300, sampler=sampler)
    prob['pstep'] = np.array(prob['pstep'], float)
    prob['pstep'][1] = 0.0
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.nfree == 2
    assert gpu.plan['persistent'] > 0 and gpu.plan['fast'] in (1, 2)
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('sampler', ['demc', 'snooker', 'mrw'])
def test_lean_instances_with_gaussian_priors(eng, sampler):
    """Gaussian priors (one symmetric, one asymmetric) and a fixed coefficient on the lean
    persistent instances: the prior term is built from the lanes in the oracle's order."""
    prob = H.quad_problem(ndata=500, nchains=64, nsamples=64*300, sampler=sampler)
    prob['pstep'] = np.array([1.0, 0.0, 0.1])
    prob['prior'] = np.array([3.2, 0.0, 0.45])
    prob['priorlow'] = np.array([0.5, 0.0, 0.02])
    prob['priorup'] = np.array([0.5, 0.0, 0.05])
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['persistent'] > 0 and gpu.plan['fast'] in (1, 2)
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('sampler', ['demc', 'snooker'])
def test_lean_instances_with_a_shared_coefficient(eng, sampler):
    """A shared parameter (pstep < 0: copies another parameter) on the lean instances."""
    x, data, uncert = H.tutorial_data(400)
    prob = dict(data=data, uncert=uncert, indep=x,
                params=np.array([3.0, -2.4, 0.5, 0.5]),          # cubic term tied to the quadratic
                pstep=np.array([1.0, 0.5, 0.1, -3.0]),
                pmin=np.array([-10.0, -20.0, -10.0, -10.0]),
                pmax=np.array([40.0, 20.0, 10.0, 10.0]),
                nchains=32, nsamples=32*300, sampler=sampler, model='poly', seed=77)
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['persistent'] > 0 and gpu.plan['fast'] in (1, 2)
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('ndata', [1500, 2048, 2049, 3000, 4096, 4097])
@pytest.mark.parametrize('wide', [True, False])
def test_lean_demc_with_more_than_one_tile(eng, monkeypatch, ndata, wide):
    """More than 1024 points on the lean DEMC loop: the tiles in the registers of four waves
    (lean_demc_wide_kernel) -- or, behind MC3HIP_NO_WIDE and for ensembles that are not
    co-resident that way, two 1024-point tiles in the registers of ONE wave and the rest streamed
    (<= 4096 points), beyond that the general instance on four waves."""
    if not wide:
        monkeypatch.setenv('MC3HIP_NO_WIDE', '1')
    prob = H.quad_problem(ndata=ndata, nchains=32, nsamples=32*150, sampler='demc')
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['persistent'] > 0
    assert (gpu.plan['fast'] == 2) == (wide or ndata <= 4096)
    assert gpu.plan['waves_per_chain'] == (4 if wide or ndata > 4096 else 1), gpu.plan
    assert_same(orc, gpu, trace=False)
    gpu.close()


def test_headline_config_c2_matches_oracle(eng):
    """BASELINE config #2 (256 chains, ndata=1000, demc) for 400 iterations:
    bit-exact against the oracle, then properties at full length."""
    prob = H.quad_problem(ndata=1000, nchains=256, nsamples=256*400, sampler='demc')
    orc, gpu = run_pair(eng, prob, trace=False)
    assert_same(orc, gpu, trace=False)
    gpu.close()


def test_headline_config_c2_properties_full_size(eng):
    """Size-independent properties on a long C2 run (20000 iterations): every
    row written once by its chain, log_post consistent with a re-evaluation of
    the stored parameters, acceptance counts consistent with the rows."""
    prob = H.quad_problem(ndata=1000, nchains=256, nsamples=256*20000, sampler='demc')
    gpu = eng.DeviceChains(**prob)
    orc = H.make_oracle(dict(prob, nsamples=256*10))
    Z0, lp0 = H.initial_population(orc)
    gpu.set_population(Z0, lp0)
    gpu.run()
    Z, lp, zc = gpu.rows()
    st = gpu.state()
    M0, n = gpu.M0, gpu.nchains
    assert np.all(zc[:M0] == -1)
    assert np.all(zc[M0:] == np.tile(np.arange(n), gpu.nzchain))
    # rows only change on acceptance: count state changes per chain
    hist = Z[M0:].reshape(gpu.nzchain, n, gpu.nfree)
    prev = np.concatenate([Z0[None, :n], hist[:-1]])
    changed = np.any(hist != prev, axis=2).sum(axis=0)
    np.testing.assert_array_equal(changed, st['numaccept'])
    # log_post of a sample of rows == -chisq/2 of the stored parameters
    idx = np.random.RandomState(0).randint(M0, len(Z), 4000)
    re = gpu.eval_chisq(Z[idx])
    np.testing.assert_array_equal(lp[idx], -0.5*re)
    # the sampler found the posterior: mean within 5 sigma/sqrt(n_eff) of truth
    post = Z[M0 + n*2000:]
    assert np.all(np.abs(post.mean(axis=0) - [3.0, -2.4, 0.5]) < [1.5, 0.6, 0.06])
    rate = st['numaccept'].sum()/(gpu.nzchain*n)
    assert 0.1 < rate < 0.5
    gpu.close()


def test_error_paths(eng):
    prob = H.quad_problem()
    with pytest.raises(Exception, match='demc needs at least 3 chains'):
        eng.DeviceChains(**dict(prob, nchains=2))
    with pytest.raises(Exception, match='power-of-two'):
        eng.DeviceChains(**dict(H.wavelet_problem(), data=np.ones(60),
                                uncert=np.ones(60), indep=np.ones(60)))
    g = eng.DeviceChains(**prob)
    with pytest.raises(Exception, match='set_population'):
        g.run(1)
    g.close()


# ---- BASELINE.json configs 4 and 5 at full size ----------------------------
def test_config_c4_wavelet_full_size(eng):
    """C4: wavelet likelihood, ndata = 2^14, 256 chains, snooker -- 12 iterations
    bit-exact against the oracle (1 chain per CU: 8 waves, 32 points per thread)."""
    prob = H.wavelet_problem(ndata=16384, nchains=256, nsamples=256*12)
    orc, gpu = run_pair(eng, prob, trace=True)
    assert gpu.plan['waves_per_chain'] == 8
    assert_same(orc, gpu)
    gpu.close()


def test_config_c5_linear_full_size(eng):
    """C5 (single-GPU share and the whole ensemble): 10-parameter linear model,
    ndata = 1e5, Gaussian priors on three parameters, bounds on all, demc."""
    for nchains, chain0, nlocal in ((512, 0, 512), (512, 128, 128)):
        prob = H.linear_problem(ndata=100000, nchains=nchains, nsamples=nchains*6)
        gpu = eng.DeviceChains(chain0=chain0, nlocal=nlocal, **prob)
        assert gpu.plan['tile'] == 2048                 # the tiled plan (10 coefficients)
        orc = H.make_oracle(prob, plan=gpu.plan, chain0=chain0, nlocal=nlocal)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        nref = 6 if nlocal == nchains else 1      # a shard can only do one refresh alone
        orc.run(nrefresh=nref, nthreads=4)
        gpu.run(nref)
        Z, lp, zc = gpu.rows()
        st = gpu.state()
        rows = slice(orc.M0, orc.M0 + nref*nchains)
        own = np.tile((np.arange(nchains) >= chain0) & (np.arange(nchains) < chain0 + nlocal), nref)
        np.testing.assert_array_equal(Z[rows][own], orc.Z[rows][own])
        np.testing.assert_array_equal(lp[rows][own], orc.log_post[rows][own])
        np.testing.assert_array_equal(st['numaccept'], orc.numaccept[chain0:chain0 + nlocal])
        np.testing.assert_array_equal(st['outbounds'], orc.outbounds)
        gpu.close()


# ---- edge cases: smallest / ragged / largest shapes -------------------------
@pytest.mark.parametrize('ndata', [1, 2, 63, 65, 1023, 1024, 1025, 2049, 5000])
@pytest.mark.parametrize('wpc', ['1', '4'])
def test_ragged_data_sizes(eng, ndata, wpc):
    """Data lengths around the 64-lane and 1024-point tile edges (single tile,
    exactly one tile, one point into the second tile, several tiles)."""
    prob = H.quad_problem(ndata=max(ndata, 1), nchains=5, nsamples=5*30, sampler='demc')
    if ndata == 1:
        prob.update(data=prob['data'][:1], uncert=np.array([1.0]), indep=prob['indep'][:1])
    os.environ['MC3HIP_WPC'] = wpc
    try:
        orc, gpu = run_pair(eng, prob)
    finally:
        del os.environ['MC3HIP_WPC']
    assert_same(orc, gpu)
    gpu.close()


def test_smallest_ensembles_and_single_free_parameter(eng):
    # demc needs 3 chains (chain.py:223-229); snooker/mrw run with 1
    for sampler, nchains in (('demc', 3), ('snooker', 1), ('mrw', 1), ('snooker', 2)):
        prob = H.quad_problem(nchains=nchains, nsamples=nchains*200, sampler=sampler,
                              hsize=10)
        orc, gpu = run_pair(eng, prob)
        assert_same(orc, gpu)
        gpu.close()
    # one free parameter: snooker's Hastings exponent (nfree-1)/2 is 0
    prob = H.quad_problem(nchains=6, nsamples=6*300, sampler='snooker',
                          pstep=np.array([1.0, 0.0, 0.0]),
                          params=np.array([3.0, -2.4, 0.5]))
    orc, gpu = run_pair(eng, prob)
    assert_same(orc, gpu)
    gpu.close()


@pytest.mark.parametrize('sampler', ['demc', 'snooker'])
def test_largest_parameter_counts(eng, sampler):
    """MC3HIP_MAX_FREE = 64 free parameters out of MC3HIP_MAX_PARAMS = 128 (the
    rest shared with a free one or fixed): a 128-column linear model."""
    rs = np.random.RandomState(3)
    ndata, npar = 300, 128
    A = rs.normal(0, 1, (npar, ndata))
    p_true = rs.normal(0, 1, npar)
    pstep = np.zeros(npar)
    pstep[:64] = 0.02
    pstep[64:96] = -(np.arange(32) + 1)        # shared with parameters 0..31
    p_true[64:96] = p_true[0:32]
    data = p_true @ A + rs.normal(0, 1, ndata)
    prior = np.zeros(npar)
    plow = np.zeros(npar)
    pup = np.zeros(npar)
    prior[10:20] = p_true[10:20]
    plow[10:20] = 0.3
    pup[10:20] = 0.6
    prob = dict(data=data, uncert=np.ones(ndata), indep=A, params=p_true.copy(),
                pstep=pstep, pmin=np.full(npar, -6.0), pmax=np.full(npar, 6.0),
                prior=prior, priorlow=plow, priorup=pup, nchains=70,
                nsamples=70*40, sampler=sampler, model='linear', seed=11)
    orc, gpu = run_pair(eng, prob)
    assert_same(orc, gpu)
    gpu.close()
    from mc3_amd import engine
    with pytest.raises(Exception, match='nfree=65 outside'):
        ps = pstep.copy()
        ps[96] = 0.02
        engine.DeviceChains(**dict(prob, pstep=ps))
    with pytest.raises(Exception, match='nparams=129 outside'):
        engine.DeviceChains(**dict(prob, params=np.zeros(129), pstep=np.ones(129),
                                   pmin=None, pmax=None, prior=None, priorlow=None,
                                   priorup=None, indep=np.zeros((129, ndata))))


def test_nonfinite_models_are_rejected_like_the_reference(eng):
    """chain.py:322-323 / SURVEY 3.3-6: a +inf model value means chisq = inf
    (always rejected); a NaN model gives a NaN chisq, and NaN > u is False."""
    x = np.linspace(0, 10, 50)
    x[7] = np.inf                                   # p1*inf -> +inf for p1 > 0
    data = np.ones(50)
    base = dict(data=data, uncert=np.ones(50), indep=x, params=np.array([1.0, 0.5]),
                pstep=np.array([0.1, 0.1]), nchains=4, nsamples=40, sampler='demc')
    with eng.DeviceChains(**base) as g:
        c = g.eval_chisq(np.array([[1.0, 0.5], [1.0, -0.5], [1.0, 0.0]]))
        orc = H.make_oracle(base).eval_chisq(np.array([[1.0, 0.5], [1.0, -0.5], [1.0, 0.0]]))
    assert c[0] == np.inf                           # +inf trapped
    assert c[1] == np.inf                           # -inf model: (m-d)^2 = inf
    assert np.isnan(c[2])                           # 0*inf = NaN model
    np.testing.assert_array_equal(c, orc)


@pytest.mark.parametrize('kind,sampler,thinning', [
    ('poly', 'demc', 1), ('poly', 'snooker', 2), ('linear', 'demc', 3),
    ('linear', 'snooker', 1)])
def test_tiled_plan_bit_exact(eng, kind, sampler, thinning):
    """Large ndata takes the tiled plan (proposal / per-tile data sums for all
    chains / accept as separate launches): 20 000 points = 5 tiles of 4096, with
    priors, bounds, a shared and a fixed parameter."""
    if kind == 'poly':
        prob = H.quad_problem(ndata=20000, nchains=40, nsamples=40*25*thinning,
                              sampler=sampler, thinning=thinning,
                              prior=np.array([3.0, 0.0, 0.5]),
                              priorlow=np.array([0.5, 0.0, 0.02]),
                              priorup=np.array([0.25, 0.0, 0.04]))
    else:
        prob = H.linear_problem(ndata=20000, nmodel=7, nchains=40,
                                nsamples=40*25*thinning, sampler=sampler,
                                thinning=thinning)
        prob['pstep'] = prob['pstep'].copy()
        prob['pstep'][5] = -2          # shared with parameter 1
        prob['pstep'][6] = 0.0         # fixed
        prob['params'] = prob['params'].copy()
        prob['params'][5] = prob['params'][1]
    orc, gpu = run_pair(eng, prob)
    assert gpu.plan['tile'] == (4096 if kind == 'poly' else 2048)
    # (one persistent launch, tests/test_gpu_resident.py; MC3HIP_NO_RESIDENT: proposal / tile
    # sums / finish as three launches per iteration)
    assert gpu.plan['persistent'] > 0 and gpu.plan['resident'] == 1
    assert_same(orc, gpu)
    os.environ['MC3HIP_NO_RESIDENT'] = '1'
    try:
        orc3, gpu3 = run_pair(eng, prob)
        assert gpu3.plan['persistent'] == 0 and gpu3.plan['resident'] == 0
        assert_same(orc3, gpu3)
        gpu3.close()
    finally:
        del os.environ['MC3HIP_NO_RESIDENT']
    # the fused plan (one workgroup streams all the data for its chain) gives the
    # same chains up to the summation order: compare chisq of stored rows
    Z, lp, zc = gpu.rows()
    gpu.close()
    os.environ['MC3HIP_NO_TILED'] = '1'
    try:
        fused = eng.DeviceChains(**prob)
        assert fused.plan['tile'] == 0
        re = fused.eval_chisq(fused.expand_free(Z[-40:]))
        fused.close()
    finally:
        del os.environ['MC3HIP_NO_TILED']
    np.testing.assert_allclose(-0.5*re, lp[-40:], rtol=1e-12)


# ---- mc3.stats' likelihood functions on arbitrary model arrays ---------------
def test_stats_functions_known_answers_of_the_reference():
    """tests/test_stats.py:115-171 of the reference, through mc3_amd.stats (device)."""
    from mc3_amd import stats as ms
    data = np.array([1.1, 1.2, 0.9, 1.0])
    model = np.ones(4)
    uncert = np.full(4, 0.1)
    np.testing.assert_allclose(ms.residuals(model, data, uncert), [-1.0, -2.0, 1.0, 0.0],
                               atol=1e-14)
    params, priors = np.array([2.5, 5.5]), np.array([2.0, 5.0])
    plow = pup = np.array([0.0, 1.0])
    np.testing.assert_allclose(ms.residuals(model, data, uncert, params, priors, plow, pup),
                               [-1.0, -2.0, 1.0, 0.0, 0.5], atol=1e-14)
    np.testing.assert_allclose(ms.chisq(model, data, uncert), 6.0, rtol=1e-14)
    np.testing.assert_allclose(ms.chisq(model, data, uncert, params, priors, plow, pup), 6.25,
                               rtol=1e-14)
    data = np.array([2.0, 0.0, 3.0, -2.0, -1.0, 2.0, 2.0, 0.0])
    model = np.ones(8)
    params = np.array([1.0, 0.1, 0.1])
    np.testing.assert_allclose(ms.dwt_chisq(model, data, params), 1693.22308882)
    np.testing.assert_allclose(
        ms.dwt_chisq(model, data, params, np.array([1.0, 0.2, 0.3]),
                     np.array([0.0, 0.0, 0.1]), np.array([0.0, 0.0, 0.1])),
        1697.2230888243134, rtol=1e-13)
    with pytest.raises(ValueError, match='at least three parameters'):
        ms.dwt_chisq(model, data, np.array([1.0, 0.1]))
    e4 = np.zeros(32)
    e4[4] = 1.0
    np.testing.assert_allclose(ms.dwt_daub4(ms.dwt_daub4(e4, True)), e4, atol=1e-8)


def test_stats_functions_vs_reference_vectors(golden_dir):
    from mc3_amd import stats as ms
    kv = np.load(os.path.join(golden_dir, 'kernel_vectors.npz'))
    for ndata in (8, 32, 1000, 16384):
        x, data, uncert = H.tutorial_data(ndata, 1000 + ndata)
        P = kv[f'gauss_{ndata}_params']
        pri = (kv['gauss_prior'], kv['gauss_priorlow'], kv['gauss_priorup'])
        got0 = [ms.chisq(H.quad(p, x), data, uncert) for p in P[:3]]
        got1 = [ms.chisq(H.quad(p, x), data, uncert, p, *pri) for p in P[:3]]
        np.testing.assert_allclose(got0, kv[f'gauss_{ndata}_chisq'][:3], rtol=1e-10)
        np.testing.assert_allclose(got1, kv[f'gauss_{ndata}_chisq_priors'][:3], rtol=1e-10)
        res = ms.residuals(H.quad(P[0], x), data, uncert, P[0], *pri)
        np.testing.assert_allclose(res, kv[f'gauss_{ndata}_residuals0'], rtol=1e-13)
    for ndata in (2, 4, 8, 32, 1024, 16384):
        rs2 = np.random.RandomState(2000 + ndata)
        x = np.linspace(0, 10, ndata)
        data = H.quad([3.0, -2.4, 0.5], x) + rs2.normal(0, 1.0, ndata) + 0.5*np.sin(2*np.pi*x)
        P = kv[f'dwt_{ndata}_params']
        pri = (kv['dwt_prior'], kv['dwt_priorlow'], kv['dwt_priorup'])
        got0 = [ms.dwt_chisq(H.quad(p[:3], x), data, p) for p in P[:3]]
        got1 = [ms.dwt_chisq(H.quad(p[:3], x), data, p, *pri) for p in P[:3]]
        np.testing.assert_allclose(got0, kv[f'dwt_{ndata}_chisq'][:3], rtol=1e-10)
        np.testing.assert_allclose(got1, kv[f'dwt_{ndata}_chisq_priors'][:3], rtol=1e-10)


@pytest.mark.parametrize('sampler', ['demc', 'snooker'])
def test_persistent_launch_gives_up_and_run_is_redone(eng, sampler):
    """If a persistent launch cannot complete (here: chain 0 gives up at refresh 7 of
    the second run() call through a test hook; in the field: workgroups not all
    resident), the library restores the checkpoint and redoes the work with one
    launch per refresh -- same results, context demoted to that path."""
    prob = H.quad_problem(nchains=12, nsamples=12*90, sampler=sampler)
    orc = H.make_oracle(prob)
    Z0, lp0 = H.initial_population(orc)
    orc.set_population(Z0, lp0)
    orc.run()
    os.environ['MC3HIP_TEST_ABORT_AT'] = '37'
    try:
        gpu = eng.DeviceChains(**prob)
    finally:
        del os.environ['MC3HIP_TEST_ABORT_AT']
    gpu.set_population(Z0, lp0)
    assert gpu.plan['persistent'] > 0
    gpu.run(30)                      # completes
    gpu.run(40, sync=False)          # gives up inside (refresh 37) ...
    gpu.run(20)                      # ... queued behind it; everything since refresh 30 is redone
    assert gpu.plan['persistent'] == 0
    Z, lp, zc = gpu.rows()
    st = gpu.state()
    np.testing.assert_array_equal(Z, orc.Z)
    np.testing.assert_array_equal(lp, orc.log_post)
    np.testing.assert_array_equal(zc, orc.zchain)
    np.testing.assert_array_equal(st['numaccept'], orc.numaccept)
    np.testing.assert_array_equal(st['outbounds'], orc.outbounds)
    np.testing.assert_array_equal(st['best_chisq'], orc.best_chisq)
    gpu.close()


@pytest.mark.parametrize('ndata,want', [(1025, 4), (2048, 4), (2049, 4), (3000, 4), (4096, 4),
                                        (4097, 4), (6000, 4), (8192, 4), (8193, 8), (12000, 8),
                                        (16384, 8)])
@pytest.mark.parametrize('case', ['plain', 'tight', 'resume', 'snooker', 'snooker-tight',
                                  'snooker-resume'])
def test_lean_loops_with_the_tiles_in_the_registers_of_several_waves(eng, monkeypatch, ndata, want,
                                                                     case):
    """1 025 ... 16 384 points of the plain strict DEMC and snooker shapes with few chains
    (lean_demc_wide_kernel / lean_snooker_wide_kernel NT: four waves keep up to eight 1024-point
    tiles between them, eight waves up to sixteen -- where one wave streamed the tiles beyond
    its second from L2, the general chain took over from 4097 points and the tiled plan from
    16 384): a virtual lane adds
    up its points tile after tile, so the sum is the canonical one bit for bit -- against the
    oracle, with bounds that reject, across launches, and behind the switch the kernels these
    shapes ran before (the same results)."""
    nref = 60
    sampler = 'snooker' if case.startswith('snooker') else 'demc'
    case = case.replace('snooker-', '').replace('snooker', 'plain')
    prob = H.quad_problem(ndata=ndata, nchains=6, nsamples=6*nref, sampler=sampler)
    if case == 'tight':
        prob.update(pmin=np.array([9.2, -2.4, 0.08]), pmax=np.array([10.8, -1.6, 0.12]))
    if case == 'resume':
        monkeypatch.setenv('MC3HIP_PERSIST_REFS', '17')
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 3, 30, 26):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert plan['fast'] == 2 and plan['persistent'] > 0 and plan['tile'] == 0, plan
    assert plan['waves_per_chain'] == want, plan
    assert_same(orc, gpu, trace=False)
    if case == 'tight':
        assert gpu.state()['outbounds'].sum() > 20
    gpu.close()
    # behind the switch: what these shapes ran before
    if case == 'plain':
        monkeypatch.setenv('MC3HIP_NO_WIDE', '1')
        orc2, gpu2 = run_pair(eng, prob, trace=False)
        assert_same(orc2, gpu2, trace=False)
        np.testing.assert_array_equal(orc2.Z, orc.Z)
        gpu2.close()


def test_wide_tiles_give_up_and_ensembles_that_do_not_fit(eng, monkeypatch):
    """The wide loop through a give-up (redone from the checkpoint, one launch per refresh), and
    an ensemble that is not co-resident at four waves per chain: back to what it ran before."""
    prob = H.quad_problem(ndata=5000, nchains=6, nsamples=6*60, sampler='demc')
    monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '23')
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['recoveries'] == 1, gpu.plan
    assert_same(orc, gpu, trace=False)
    gpu.close()
    monkeypatch.delenv('MC3HIP_TEST_ABORT_AT')
    monkeypatch.setenv('MC3HIP_PERSIST_CAPACITY', '4')
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['fast'] != 2 or gpu.plan['chains_per_workgroup'] > 1, gpu.plan
    assert_same(orc, gpu, trace=False)
    gpu.close()


# ---- wavelet likelihood: the hand-structured chain wave (kernels_wavelet.h) -------------
def _cw_problem(sampler, ndata, case):
    prob = H.wavelet_problem(ndata=ndata, nchains=12, nsamples=12*160, sampler=sampler)
    if case == 'epsilon':
        prob.update(fepsilon=0.03, fgamma=0.8)
    elif case == 'tight':           # most proposals leave the box: the likelihood is skipped
        prob.update(pmin=np.array([2.6, -2.6, 0.47, 0.0, 0.3, 0.8]),
                    pmax=np.array([3.4, -2.2, 0.53, 2.0, 0.7, 1.2]))
    elif case == 'wide':            # none does
        prob.update(pmin=np.full(6, -1e3), pmax=np.full(6, 1e3))
    return prob


@pytest.mark.parametrize('case', ['plain', 'epsilon', 'tight', 'wide'])
@pytest.mark.parametrize('ndata', [8192, 16384])
@pytest.mark.parametrize('sampler', ['snooker', 'demc', 'mrw'])
def test_wavelet_lean_chain_wave_bit_exact(eng, sampler, ndata, case):
    """Without trace / priors / shared parameters / thinning the ninth wave of the wavelet
    kernel runs lean_chain_wave instead of the general run_chain: same rows, counters, best
    fits as the oracle (and therefore as the general chain, which the traced tests run)."""
    orc, gpu = run_pair(eng, _cw_problem(sampler, ndata, case), trace=False)
    plan = gpu.plan
    assert plan['fast'] == 3 and plan['persistent'] > 0 and plan['recoveries'] == 0, plan
    assert_same(orc, gpu, trace=False)
    if case == 'tight':
        assert gpu.state()['outbounds'].sum() > 100
    gpu.close()


@pytest.mark.parametrize('sampler', ['snooker', 'demc', 'mrw'])
def test_wavelet_lean_chain_wave_resume_abort_and_switch(eng, monkeypatch, sampler):
    prob = _cw_problem(sampler, 16384, 'plain')
    # several launches: every launch starts without anything prepared ahead
    gpu = eng.DeviceChains(**prob)
    orc = H.make_oracle(prob, plan=gpu.plan)
    Z0, lp0 = H.initial_population(orc)
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(nthreads=4)
    for n in (1, 2, 64, 65, 28):
        gpu.run(n)
    assert gpu.plan['fast'] == 3
    assert_same(orc, gpu, trace=False)
    gpu.close()
    # a launch that gives up (test hook) is redone from the checkpoint
    monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '21')
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['recoveries'] == 1
    assert_same(orc, gpu, trace=False)
    gpu.close()
    monkeypatch.delenv('MC3HIP_TEST_ABORT_AT')
    # the general chain behind the same kernel
    monkeypatch.setenv('MC3HIP_NO_CW_LEAN', '1')
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['fast'] == 0 and gpu.plan['persistent'] > 0
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('case', ['quad', 'tight', 'ragged', 'resume'])
def test_lean_demc_with_the_draws_of_64_refreshes_at_once(eng, case):
    """lean_demc_kernel BATCH (the plain shape's loop at every size since round 6): Philox, log u
    and the partners of 64 consecutive refreshes computed at once, one refresh per lane; launches
    that start and stop inside a batch."""
    if case == 'ragged':
        prob = H.quad_problem(ndata=37, nchains=9, nsamples=9*400)
    elif case == 'tight':
        prob = H.quad_problem(ndata=200, nchains=12, nsamples=12*400)
        prob.update(params=np.array([4.0, -2.5, 0.5]), pmin=np.array([3.5, -2.8, 0.45]),
                    pmax=np.array([6.5, -2.2, 0.56]))
    else:
        prob = H.quad_problem(ndata=1000, nchains=40, nsamples=40*300)
    if case == 'resume':
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 64, 65, 100, 68):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert plan['fast'] == 2 and plan['persistent'] > 0 and plan['recoveries'] == 0, plan
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('case', ['launches', 'giveup'])
def test_lean_demc_big_ensemble_wait_laid_out_by_hand(eng, monkeypatch, case):
    """lean_demc_kernel BIG (more than 1 024 chains: kernels_lean.h): the wait for the partners'
    rows with its own loop -- launches that start and stop inside a batch of draws, and a launch
    that gives up (test hook: chain 0 raises the flag at refresh 21, every other chain finds it
    from inside its wait) and is redone from the checkpoint."""
    prob = H.quad_problem(ndata=100, nchains=1100, nsamples=1100*200)
    if case == 'giveup':
        monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '21')
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['recoveries'] == 1, gpu.plan
    else:
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=8)
        for n in (1, 2, 64, 65, 68):
            gpu.run(n)
        assert gpu.plan['recoveries'] == 0, gpu.plan
        kernel = gpu.last_kernel
        assert gpu.plan['fast'] == 2 and 'lean_demc_kernel' in kernel, (gpu.plan, kernel)
        # (the last template argument, mangled or not)
        assert 'Lb1EEEvNS' in kernel or ', true>(' in kernel, kernel
    # (after a launch that gave up the context goes on a launch per refresh: its plan says so)
    assert_same(orc, gpu, trace=False)
    gpu.close()


# ---- wavelet likelihood x design-matrix model ---------------------------------------------
@pytest.mark.parametrize('trace', [True, False])
@pytest.mark.parametrize('ndata,nmodel,sampler', [
    (64, 4, 'snooker'), (512, 3, 'demc'), (2048, 6, 'mrw'), (4096, 5, 'snooker'),
    (8192, 4, 'demc'), (16384, 7, 'snooker'), (16384, 3, 'mrw')])
def test_wavelet_likelihood_with_a_design_matrix_model(eng, ndata, nmodel, sampler, trace):
    """Every wavelet geometry (one, four, eight waves; traced: the general chain, untraced at
    8192 / 16384 points: the nine-wave kernel with the hand-structured chain wave) with the
    model values formed column by column (wavelet_level0, MT_LINEAR)."""
    prob = H.wavelet_linear_problem(ndata=ndata, nmodel=nmodel, nchains=7, nsamples=7*24,
                                    sampler=sampler)
    orc, gpu = run_pair(eng, prob, trace=trace)
    assert gpu.plan['persistent'] > 0 and gpu.plan['recoveries'] == 0
    if ndata >= 8192 and not trace:
        assert gpu.plan['fast'] == 3
    assert_same(orc, gpu, trace=trace)
    from oracle import kernels as ok
    P = prob['params'] + np.random.RandomState(2).normal(0, 1, (6, nmodel + 3))*prob['pstep']
    got = gpu.eval_chisq(P)
    want = [ok.dwt_chisq(p[:nmodel] @ prob['indep'], prob['data'], p) for p in P]
    np.testing.assert_allclose(got, want, rtol=1e-10)
    gpu.close()


@pytest.mark.parametrize('case', ['quad', 'line', 'cubic', 'tight', 'ragged', 'resume', 'abort',
                                  'switch', 'fixed', 'priors'])
def test_lean_mrw_loop(eng, monkeypatch, case):
    """lean_mrw_kernel: Metropolis draws, their log windows and the normal deviates of 21
    refreshes at once; two, three and four parameters (one or two slots of deviates)."""
    if case == 'ragged':
        prob = H.quad_problem(ndata=37, nchains=9, nsamples=9*400, sampler='mrw')
    elif case == 'tight':
        prob = H.quad_problem(ndata=200, nchains=12, nsamples=12*400, sampler='mrw')
        prob.update(params=np.array([4.0, -2.5, 0.5]), pmin=np.array([3.5, -2.8, 0.45]),
                    pmax=np.array([6.5, -2.2, 0.56]), pstep=np.array([0.5, 0.2, 0.05]))
    elif case in ('line', 'cubic'):
        ncoef = 2 if case == 'line' else 4
        x, data, uncert = H.tutorial_data(600)
        prob = dict(data=data, uncert=uncert, indep=x/10.0,
                    params=np.array([3.0, -20.0, 40.0, 5.0][:ncoef]),
                    pstep=np.array([0.05, 0.1, 0.2, 0.1][:ncoef]), nchains=16,
                    nsamples=16*300, sampler='mrw', model='poly', seed=9)
    else:
        prob = H.quad_problem(ndata=1000, nchains=40, nsamples=40*300, sampler='mrw')
        prob['pstep'] = np.array([0.05, 0.02, 0.003])       # (steps that do get accepted)
    if case == 'fixed':                    # (FX instance: the linear coefficient is fixed)
        prob['pstep'] = np.array(prob['pstep'], dtype=float)
        prob['pstep'][1] = 0.0
    if case == 'priors':                   # (FX instance: a Gaussian prior, a shared coefficient)
        prob.update(prior=np.array([4.0, 0.0, 0.0]), priorlow=np.array([0.5, 0.0, 0.0]),
                    priorup=np.array([0.7, 0.0, 0.0]))
    if case == 'switch':
        monkeypatch.setenv('MC3HIP_NO_LEAN', '1')
    if case == 'abort':
        monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '21')
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['recoveries'] == 1
        assert_same(orc, gpu, trace=False)
        gpu.close()
        return
    if case == 'resume':
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 20, 21, 22, 64, 100, 70):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert plan['persistent'] > 0 and plan['recoveries'] == 0, plan
    assert plan['fast'] == (0 if case == 'switch' else 2), plan
    assert_same(orc, gpu, trace=False)
    if case == 'tight':
        assert gpu.state()['outbounds'].sum() > 50
    gpu.close()


@pytest.mark.parametrize('sampler', ['snooker', 'mrw', 'demc'])
@pytest.mark.parametrize('thinning', [2, 3, 7, 25])
@pytest.mark.parametrize('case', ['plain', 'resume', 'abort', 'tight', 'tiny-history'])
def test_lean_snooker_and_mrw_loops_with_thinning(eng, monkeypatch, case, thinning, sampler):
    """The THIN instances of lean_snooker_kernel / lean_mrw_kernel and lean_demc_lag_kernel (zlag 0):
    `thinning` iterations per refresh, the draws keyed by the iteration, only the last one of a refresh published; a
    batch of draws (21 iterations) ends anywhere inside a refresh."""
    nchains = 5 if case == 'tiny-history' else 12
    prob = H.quad_problem(ndata=300, nchains=nchains, nsamples=nchains*90*thinning,
                          sampler=sampler, thinning=thinning)
    if sampler == 'mrw':
        prob['pstep'] = np.array([0.05, 0.02, 0.003])
    if case == 'tight':
        prob.update(params=np.array([4.0, -2.5, 0.5]), pmin=np.array([3.5, -2.8, 0.45]),
                    pmax=np.array([6.5, -2.2, 0.56]))
    if case == 'tiny-history':
        prob.update(hsize=1)
    if case == 'abort':
        monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '21')
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['recoveries'] == 1
        assert_same(orc, gpu, trace=False)
        gpu.close()
        return
    if case == 'resume':
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 20, 21, 22, 24):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert plan['persistent'] > 0 and plan['recoveries'] == 0 and plan['fast'] == 2, plan
    assert_same(orc, gpu, trace=False)
    if case == 'tight':
        assert gpu.state()['outbounds'].sum() > 50
    gpu.close()


@pytest.mark.parametrize('sampler', ['snooker', 'mrw'])
@pytest.mark.parametrize('ndata', [1025, 2048, 2049, 4096])
def test_lean_snooker_and_mrw_loops_with_streamed_tiles(eng, monkeypatch, ndata, sampler):
    """The MULTI instances of lean_snooker_kernel / lean_mrw_kernel: two tiles in registers, the
    third and fourth streamed inside the loop (loads the compiler tracks next to the ones it
    does not).  (Snooker: what runs behind MC3HIP_NO_WIDE and for ensembles that are not
    co-resident at four waves per chain -- lean_snooker_wide_kernel otherwise.)"""
    monkeypatch.setenv('MC3HIP_NO_WIDE', '1')
    prob = H.quad_problem(ndata=ndata, nchains=14, nsamples=14*150, sampler=sampler)
    if sampler == 'mrw':
        prob['pstep'] = np.array([0.05, 0.02, 0.003])
    gpu = eng.DeviceChains(**prob)
    orc = H.make_oracle(prob, plan=gpu.plan)
    Z0, lp0 = H.initial_population(orc)
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(nthreads=4)
    for n in (1, 30, 22, 97):
        gpu.run(n)
    plan = gpu.plan
    assert plan['persistent'] > 0 and plan['recoveries'] == 0 and plan['fast'] == 2, plan
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('case', ['quad', 'line', 'cubic', 'tight', 'ragged', 'resume', 'abort',
                                  'tiny-history', 'switch', 'fixed', 'priors'])
def test_lean_snooker_loop(eng, monkeypatch, case):
    """lean_snooker_kernel: the draws of 21 refreshes at once with their row offsets, jump
    flags and log-u windows; rows asked for a refresh ahead; ordinary jumps decided in the log
    domain, snooker jumps proper (one in ten) with the Hastings factor.  A short initial
    history (hsize 1: the newest rows are drawn often and are still pending when asked for)."""
    if case == 'ragged':
        prob = H.quad_problem(ndata=37, nchains=9, nsamples=9*400, sampler='snooker')
    elif case == 'tight':
        prob = H.quad_problem(ndata=200, nchains=12, nsamples=12*400, sampler='snooker')
        prob.update(params=np.array([4.0, -2.5, 0.5]), pmin=np.array([3.5, -2.8, 0.45]),
                    pmax=np.array([6.5, -2.2, 0.56]))
    elif case in ('line', 'cubic'):
        ncoef = 2 if case == 'line' else 4
        x, data, uncert = H.tutorial_data(600)
        prob = dict(data=data, uncert=uncert, indep=x/10.0,
                    params=np.array([3.0, -20.0, 40.0, 5.0][:ncoef]),
                    pstep=np.array([0.5, 1.0, 2.0, 1.0][:ncoef]), nchains=16,
                    nsamples=16*300, sampler='snooker', model='poly', seed=9)
    elif case == 'tiny-history':
        prob = H.quad_problem(ndata=300, nchains=5, nsamples=5*500, sampler='snooker', hsize=1)
    else:
        prob = H.quad_problem(ndata=1000, nchains=40, nsamples=40*300, sampler='snooker')
    if case == 'fixed':                    # (FX instance: the linear coefficient is fixed)
        prob['pstep'] = np.array(prob['pstep'], dtype=float)
        prob['pstep'][1] = 0.0
    if case == 'priors':                   # (FX instance: a Gaussian prior, a shared coefficient)
        prob.update(prior=np.array([4.0, 0.0, 0.0]), priorlow=np.array([0.5, 0.0, 0.0]),
                    priorup=np.array([0.7, 0.0, 0.0]))
    if case == 'switch':
        monkeypatch.setenv('MC3HIP_NO_LEAN', '1')
    if case == 'abort':
        monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '21')
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['recoveries'] == 1
        assert_same(orc, gpu, trace=False)
        gpu.close()
        return
    if case == 'resume':
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 20, 21, 22, 64, 100, 70):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert plan['persistent'] > 0 and plan['recoveries'] == 0, plan
    assert plan['fast'] == (0 if case == 'switch' else 2), plan
    assert_same(orc, gpu, trace=False)
    if case == 'tight':
        assert gpu.state()['outbounds'].sum() > 50
    gpu.close()
