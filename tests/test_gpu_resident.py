"""The tiled plan in ONE persistent launch (resident_tile_kernel, resident.h): every
workgroup keeps its tile of the design matrix on chip, proposals and per-tile sums are
handed over inside the launch.  Same arithmetic as the three-launch tiled plan, so the same
oracle (oracle/oracle_lockstep.c, `tile_tree = 2`) checks both, bit for bit.

Reference path: mc3/chain.py:185-289 with the model call of chain.py:317-319 (a design
matrix times the parameter vector) and the Gaussian chisq of src_c/_chisq.c:111-140.
"""
import os

import numpy as np
import pytest

import helpers as H
from test_gpu_parity import run_pair, assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from mc3_amd import engine
    return engine


class env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update({k: str(v) for k, v in self.kw.items()})

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def linear(ndata=20000, nmodel=7, nchains=40, niter=25, sampler='demc', thinning=1, **kw):
    return H.linear_problem(ndata=ndata, nmodel=nmodel, nchains=nchains,
                            nsamples=nchains*niter*thinning, sampler=sampler,
                            thinning=thinning, **kw)


@pytest.mark.parametrize('nmodel,sampler,thinning', [
    (7, 'demc', 1), (7, 'snooker', 1), (7, 'mrw', 2), (10, 'demc', 3), (10, 'snooker', 2),
    (5, 'demc', 1), (11, 'snooker', 1), (8, 'demc', 1)])
def test_resident_plan_bit_exact(eng, nmodel, sampler, thinning):
    """20 000 points = 10 tiles of 2048; a shared and a fixed parameter, priors, bounds;
    5..11 coefficients = two or three blocks of operand rows (the last one in LDS), the data
    column riding as operand row K."""
    prob = linear(nmodel=nmodel, sampler=sampler, thinning=thinning)
    prob['pstep'] = prob['pstep'].copy()
    prob['params'] = prob['params'].copy()
    prob['pstep'][nmodel - 2] = -2          # shared with parameter 1
    prob['pstep'][nmodel - 1] = 0.0         # fixed
    prob['params'][nmodel - 2] = prob['params'][1]
    orc, gpu = run_pair(eng, prob)
    assert gpu.plan['tile'] == 2048 and gpu.plan['tile_tree'] == 2
    assert gpu.plan['resident'] == 1 and gpu.plan['persistent'] > 0
    assert_same(orc, gpu)
    assert gpu.plan['recoveries'] == 0
    gpu.close()


def test_three_launch_plan_still_there(eng):
    """MC3HIP_NO_RESIDENT=1: the same problem through propose / tile sums / finish as
    separate launches gives the same chains (it is what a recovery falls back to)."""
    prob = linear(nmodel=7, sampler='snooker')
    with env(MC3HIP_NO_RESIDENT=1):
        orc, gpu = run_pair(eng, prob)
        assert gpu.plan['resident'] == 0 and gpu.plan['persistent'] == 0
        assert_same(orc, gpu)
        gpu.close()


@pytest.mark.parametrize('nmodel,sampler', [(2, 'demc'), (3, 'snooker'), (4, 'demc'), (12, 'demc'),
                                            (13, 'snooker'), (15, 'demc'), (15, 'mrw')])
def test_resident_plan_from_two_to_fifteen_columns(eng, nmodel, sampler):
    """Design matrices of 2..4 columns (one or two blocks of operand rows) and of 12..15 (four:
    the last in LDS) take the one-launch plan too; the three-launch plan computes the same."""
    prob = linear(nmodel=nmodel, sampler=sampler, niter=12)
    orc, gpu = run_pair(eng, prob)
    assert gpu.plan['tile'] == 2048 and gpu.plan['tile_tree'] == 2, gpu.plan
    assert gpu.plan['resident'] == 1 and gpu.plan['recoveries'] == 0, gpu.plan
    assert_same(orc, gpu)
    gpu.close()
    with env(MC3HIP_NO_RESIDENT=1):
        orc, gpu = run_pair(eng, prob)
        assert gpu.plan['resident'] == 0 and gpu.plan['tile_tree'] == 2
        assert_same(orc, gpu)
        gpu.close()


def test_sixteen_coefficients_keep_the_vector_tile_kernel(eng):
    """K = 16 plus the data column would be a fifth block of operand rows: the vector tile
    kernel (three launches, lanes-first order)."""
    prob = linear(nmodel=16, niter=6)
    orc, gpu = run_pair(eng, prob)
    assert gpu.plan['tile'] == 2048 and gpu.plan['resident'] == 0 and gpu.plan['tile_tree'] == 0
    assert_same(orc, gpu)
    gpu.close()


def test_resident_several_launches_and_resume(eng):
    """Launches of 7 refreshes (MC3HIP_PERSIST_REFS), the run split over three calls: the
    hand-off areas must be back to "nothing there" at every launch boundary."""
    prob = linear(nmodel=9, nchains=48, niter=40, sampler='snooker')
    with env(MC3HIP_PERSIST_REFS=7):
        gpu = eng.DeviceChains(**prob)
        assert gpu.plan['resident'] == 1 and gpu.plan['persistent'] == 7
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nrefresh=40, trace=False, nthreads=4)
        for n in (3, 23, 14):
            gpu.run(n)
        assert_same(orc, gpu, trace=False)
        gpu.close()


def test_resident_many_out_of_bounds(eng):
    """Tight bounds: most proposals leave them, so owners consume sums they do not use
    (the proposal row may only be reused once every tile has answered)."""
    prob = linear(nmodel=6, nchains=32, niter=60)
    prob['pmin'] = prob['params'] - 1.2*np.abs(prob['pstep'])
    prob['pmax'] = prob['params'] + 1.2*np.abs(prob['pstep'])
    orc, gpu = run_pair(eng, prob)
    assert gpu.plan['resident'] == 1
    assert orc.outbounds.sum() > 50
    assert_same(orc, gpu)
    gpu.close()


def test_resident_more_than_64_tiles(eng):
    """140 000 points = 69 tiles: the owner collects the sums 64 at a time; three columns of
    chains fit the chip."""
    prob = linear(ndata=140000, nmodel=10, nchains=96, niter=5)
    orc, gpu = run_pair(eng, prob)
    assert gpu.plan['resident'] == 1
    assert_same(orc, gpu)
    gpu.close()


def test_resident_gives_up_and_is_redone(eng):
    """A wave gives up at refresh 9 (test hook): every workgroup leaves, the host restores
    its checkpoint and redoes the run with the three-launch plan; same chains."""
    prob = linear(nmodel=7, nchains=40, niter=25)
    with env(MC3HIP_TEST_ABORT_AT=9):
        orc, gpu = run_pair(eng, prob, nrefresh=20)
    assert gpu.plan['recoveries'] == 1 and gpu.plan['persistent'] == 0
    tc, tf = gpu.trace(20)
    np.testing.assert_array_equal(tf, orc.trace_flag)
    np.testing.assert_array_equal(tc, orc.trace_chisq)
    gpu.run(5)                                  # and goes on from there
    orc.run(nrefresh=5, trace=False, nthreads=4)
    assert_same(orc, gpu, trace=False)
    gpu.close()


def test_config_c5_resident(eng):
    """C5 at full size (10 coefficients, 1e5 points, 512 chains): 49 tiles x 5 columns of
    112 chains = 245 workgroups, one per CU."""
    prob = H.linear_problem(ndata=100000, nchains=512, nsamples=512*8)
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['resident'] == 1
    # (three chains to own per workgroup: the owners are waves of their own)
    assert gpu.plan['owner_waves'] == 3, gpu.plan
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('case', range(10))
def test_resident_random_problems(eng, case):
    """Random shapes of the resident plan: 5..11 coefficients, 16..90 chains (one to six
    blocks of 16 per column, ragged last block), 8..20 tiles with a ragged last one, any
    sampler, thinning, epsilon, fixed / shared parameters."""
    rs = np.random.RandomState(9000 + case)
    nmodel = int(rs.randint(5, 12))
    nchains = int(rs.randint(16, 91))
    ndata = int(rs.randint(16384, 41000))
    sampler = ['demc', 'snooker', 'mrw'][case % 3]
    thinning = int(rs.choice([1, 1, 2, 3]))
    prob = linear(ndata=ndata, nmodel=nmodel, nchains=nchains, niter=12, sampler=sampler,
                  thinning=thinning)
    prob['pstep'] = prob['pstep'].copy()
    prob['params'] = prob['params'].copy()
    if rs.rand() < 0.5:
        prob['pstep'][nmodel - 1] = 0.0
    if rs.rand() < 0.5:
        prob['pstep'][nmodel - 2] = -1          # shared with parameter 0
        prob['params'][nmodel - 2] = prob['params'][0]
    if sampler != 'mrw' and rs.rand() < 0.5:
        prob['fepsilon'] = 0.01
    orc, gpu = run_pair(eng, prob)
    assert gpu.plan['resident'] == 1, (nmodel, nchains, ndata)
    assert_same(orc, gpu)
    gpu.close()


def test_resident_column_longer_than_the_lds_window(eng):
    """One column of 320 chains (MC3HIP_RESIDENT_MAXCOL=1): the group totals of a workgroup
    are collected in LDS for 256 chains at a time, so this column is added up in two
    windows."""
    prob = linear(ndata=140000, nmodel=9, nchains=320, niter=4)
    with env(MC3HIP_RESIDENT_MAXCOL=1):
        orc, gpu = run_pair(eng, prob)
        assert gpu.plan['resident'] == 1
        assert_same(orc, gpu)
        gpu.close()


# ---- polynomials of 2..4 coefficients: the vector-unit flavour (ResidentPoly) --------------
def poly(ncoef, ndata=20000, nchains=40, niter=25, sampler='demc', thinning=1, **kw):
    """The tutorial's quadratic data fitted with a polynomial of `ncoef` coefficients."""
    prob = H.quad_problem(ndata=ndata, nchains=nchains, nsamples=nchains*niter*thinning,
                          sampler=sampler, thinning=thinning, **kw)
    params = np.array([10.0, -2.0, 0.1, 0.001])[:ncoef]
    pstep = np.array([1.0, 0.5, 0.1, 0.002])[:ncoef]
    pmin = np.array([-10.0, -20.0, -10.0, -1.0])[:ncoef]
    pmax = np.array([40.0, 20.0, 10.0, 1.0])[:ncoef]
    prob.update(params=params, pstep=pstep, pmin=pmin, pmax=pmax)
    for key in ('prior', 'priorlow', 'priorup'):
        if key in prob:
            prob[key] = np.resize(prob[key], ncoef)
    return prob


@pytest.mark.parametrize('ncoef,sampler,thinning', [
    (3, 'demc', 1), (3, 'snooker', 1), (3, 'mrw', 2), (2, 'demc', 3), (4, 'snooker', 2),
    (4, 'demc', 1)])
def test_resident_polynomial_bit_exact(eng, ncoef, sampler, thinning):
    """20 000 points = 5 tiles of 4096, 40 chains = 3 columns: proposals, Horner sums on the
    vector unit and accept in one persistent launch; priors and bounds; summation order of
    the three-launch plan (lanes first), so the same oracle."""
    prob = poly(ncoef, sampler=sampler, thinning=thinning,
                prior=np.array([3.0, 0.0, 0.5]), priorlow=np.array([0.5, 0.0, 0.02]),
                priorup=np.array([0.25, 0.0, 0.04]))
    orc, gpu = run_pair(eng, prob)
    assert gpu.plan['tile'] == 4096 and gpu.plan['tile_tree'] == 0
    assert gpu.plan['resident'] == 1 and gpu.plan['persistent'] > 0
    assert_same(orc, gpu)
    assert gpu.plan['recoveries'] == 0
    gpu.close()


def test_resident_polynomial_shared_and_fixed(eng):
    """A shared and a fixed coefficient (the general run_chain carries them)."""
    prob = poly(4, sampler='snooker')
    prob['pstep'] = prob['pstep'].copy()
    prob['params'] = prob['params'].copy()
    prob['pstep'][3] = 0.0                  # fixed
    prob['pstep'][2] = -2                   # shared with parameter 1
    prob['params'][2] = prob['params'][1]
    prob['pmin'] = prob['pmin'].copy()
    prob['pmax'] = prob['pmax'].copy()
    prob['pmin'][2], prob['pmax'][2] = prob['pmin'][1], prob['pmax'][1]
    orc, gpu = run_pair(eng, prob)
    assert gpu.plan['resident'] == 1
    assert_same(orc, gpu)
    gpu.close()


@pytest.mark.parametrize('case', range(8))
def test_resident_polynomial_random_problems(eng, case):
    rs = np.random.RandomState(7000 + case)
    ncoef = int(rs.randint(2, 5))
    nchains = int(rs.randint(16, 120))
    ndata = int(rs.randint(16384, 70000))
    sampler = ['demc', 'snooker', 'mrw'][case % 3]
    thinning = int(rs.choice([1, 1, 2]))
    kw = {'fepsilon': 0.01} if (sampler != 'mrw' and rs.rand() < 0.5) else {}
    prob = poly(ncoef, ndata=ndata, nchains=nchains, niter=10, sampler=sampler,
                thinning=thinning, **kw)
    orc, gpu = run_pair(eng, prob)
    assert gpu.plan['resident'] == 1, (ncoef, nchains, ndata)
    assert_same(orc, gpu)
    gpu.close()


def test_resident_polynomial_gives_up_and_is_redone(eng):
    prob = poly(3, niter=25)
    with env(MC3HIP_TEST_ABORT_AT=9):
        orc, gpu = run_pair(eng, prob, nrefresh=20)
    assert gpu.plan['recoveries'] == 1 and gpu.plan['persistent'] == 0
    tc, tf = gpu.trace(20)
    np.testing.assert_array_equal(tf, orc.trace_flag)
    np.testing.assert_array_equal(tc, orc.trace_chisq)
    gpu.run(5)
    orc.run(nrefresh=5, trace=False, nthreads=4)
    assert_same(orc, gpu, trace=False)
    gpu.close()


def test_resident_polynomial_large(eng):
    """512 chains x 1e5 points (25 tiles x 10 columns = 250 workgroups), the quadratic."""
    prob = H.quad_problem(ndata=100000, nchains=512, nsamples=512*6)
    orc, gpu = run_pair(eng, prob, trace=False)
    assert gpu.plan['resident'] == 1
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('kind,ndata,nchains', [('poly', 12000, 640), ('linear', 9000, 800),
                                                ('poly', 12000, 256)])
def test_large_ensembles_take_the_tiled_plan_from_8192_points(eng, kind, ndata, nchains):
    """Below 16 384 points the tiled (resident) plan is chosen when the ensemble is large
    enough to pay for its fixed cost (>= 4.2 M chain-points per iteration); the third case
    stays on the fused kernel.  Bit-exact either way: the oracle follows the plan."""
    if kind == 'poly':
        prob = poly(3, ndata=ndata, nchains=nchains, niter=6)
    else:
        prob = linear(ndata=ndata, nmodel=8, nchains=nchains, niter=6)
    orc, gpu = run_pair(eng, prob, trace=False)
    tiled = ndata*nchains >= 4194304
    assert (gpu.plan['tile'] > 0) == tiled and gpu.plan['resident'] == (1 if tiled else 0)
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('kind', ['poly', 'linear'])
def test_resident_nonfinite_models_are_rejected_like_the_reference(eng, kind):
    """chain.py:322-323 under the one-launch tiled plan: an abscissa / design value of +inf
    makes the model +inf or NaN there depending on the sign of the proposed coefficient; the
    tile sum is then non-finite and only the scan of the model values tells "+inf model"
    (chisq = inf) from any other overflow (NaN chisq) -- both rejected, counters, rows and
    per-step chisq as the oracle's."""
    if kind == 'poly':
        prob = poly(3, ndata=20000, nchains=32, niter=15)
        prob['indep'] = prob['indep'].copy()
        prob['indep'][777] = np.inf
    else:
        prob = linear(ndata=20000, nmodel=6, nchains=32, niter=15)
        prob['indep'] = prob['indep'].copy()
        prob['indep'][2, 12345] = np.inf
    gpu = eng.DeviceChains(trace_iters=15, **prob)
    assert gpu.plan['resident'] == 1
    orc = H.make_oracle(prob, plan=gpu.plan)
    # (a population that is finite in spite of the bad point cannot be drawn: start from
    # arbitrary in-bounds rows with made-up log-posteriors, the same on both sides)
    rs = np.random.RandomState(1)
    ifree = orc.ifree
    Z0 = rs.normal(orc.params[ifree], 0.3*orc.pstep[ifree], (orc.M0, orc.nfree))
    lp0 = -0.5*rs.uniform(2e4, 3e4, orc.M0)
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(trace=True, nthreads=4)
    gpu.run()
    tc, tf = gpu.trace(15)
    np.testing.assert_array_equal(tf, orc.trace_flag)
    np.testing.assert_array_equal(tc, orc.trace_chisq)          # (NaN == NaN here)
    Z, lp, zc = gpu.rows()
    np.testing.assert_array_equal(Z, orc.Z)
    np.testing.assert_array_equal(lp, orc.log_post)
    np.testing.assert_array_equal(gpu.state()['outbounds'], orc.outbounds)
    inb = (tf & 1) != 0
    assert inb.any() and not np.isfinite(tc[inb]).any()      # every in-bounds proposal: inf / NaN
    assert (np.isnan(tc[inb])).any() or (tc[inb] == np.inf).any()
    assert gpu.state()['numaccept'].sum() == 0
    gpu.close()


@pytest.mark.parametrize('plan_env,want', [
    ({}, dict(resident=1, tile_tree=2)),                          # one launch, matrix cores
    ({'MC3HIP_NO_RESIDENT': '1'}, dict(resident=0, tile_tree=2)),  # three launches, matrix cores
])
def test_tiled_plans_against_the_reference_chisq_at_c5_size(eng, monkeypatch, golden_dir,
                                                            plan_env, want):
    """An independent check of the tiled plans' data sums at BASELINE config 5's own size
    (1e5 points x 10 columns), against the REFERENCE's _chisq (tests/golden/
    residue_vectors.npz: model by NumPy, chisq by src_c/_chisq.c), not against a checker
    that follows the library's summation order.  The chain kernels are made to evaluate a
    given vector p: with every row of the population equal to p a DEMC proposal is p exactly
    (gamma (x1 - x2) = 0), and its chisq comes back in the per-step trace."""
    for k, v in plan_env.items():
        monkeypatch.setenv(k, v)
    kv = np.load(os.path.join(golden_dir, 'residue_vectors.npz'))
    nchains = 16
    prob = H.linear_problem(ndata=100000, nmodel=10, nchains=nchains, nsamples=nchains)
    np.testing.assert_allclose([prob['data'].sum(), prob['indep'].sum()],
                               kv['c5_data_checksum'], rtol=1e-12)
    for with_priors, key in ((True, 'c5_chisq_priors'), (False, 'c5_chisq')):
        p2 = dict(prob) if with_priors else dict(prob, prior=None, priorlow=None, priorup=None)
        for p, ref in zip(kv['c5_params'], kv[key]):
            with eng.DeviceChains(trace_iters=1, **p2) as gpu:
                plan = gpu.plan
                assert plan['tile'] == 2048 and plan['tile_tree'] == want['tile_tree'], plan
                assert plan['resident'] == want['resident'], plan
                Z0 = np.tile(p, (gpu.M0, 1))
                gpu.set_population(Z0, np.full(gpu.M0, -1e300), X0=Z0[:nchains],
                                   chisq0=np.full(nchains, 1e300))
                gpu.run(1)
                tc, tf = gpu.trace(1)
                Z, lp, _ = gpu.rows(gpu.M0, nchains)
            assert np.all(tf[0] == 3)                      # in bounds, accepted
            np.testing.assert_array_equal(Z, Z0[:nchains])  # the proposal was p itself
            assert np.all(tc[0] == tc[0][0])
            np.testing.assert_allclose(tc[0], ref, rtol=1e-10)
            np.testing.assert_allclose(-2.0*lp, ref, rtol=1e-10)


# ---- the hand-structured owner (resident.h lean_resident_owner) -----------------------------
def _lean_case(case):
    if case == 'c5like':                      # priors on three parameters, bounds on all
        return linear(nmodel=10, nchains=48, niter=40)
    if case == 'fixed':                       # a fixed coefficient
        prob = linear(nmodel=7, nchains=40, niter=40)
        prob['pstep'] = prob['pstep'].copy()
        prob['pstep'][5] = 0.0
        return prob
    if case == 'epsilon':
        return linear(nmodel=5, nchains=40, niter=40, fepsilon=0.02, fgamma=0.9)
    if case == 'tight':                       # most proposals leave the box
        prob = linear(nmodel=6, nchains=32, niter=60)
        prob['pmin'] = prob['params'] - 1.2*np.abs(prob['pstep'])
        prob['pmax'] = prob['params'] + 1.2*np.abs(prob['pstep'])
        return prob
    if case == 'tiles69':                     # sums collected 64 at a time
        return linear(ndata=140000, nmodel=10, nchains=96, niter=5)
    if case == 'wide12':                      # a block of operand rows in LDS
        return linear(nmodel=13, nchains=40, niter=30)
    if case == 'quad':
        return poly(3, niter=40)
    if case == 'line':
        return poly(2, niter=40)
    if case == 'cubic':
        return poly(4, niter=40, fepsilon=0.01)
    raise ValueError(case)


@pytest.mark.parametrize('case', ['c5like', 'fixed', 'epsilon', 'tight', 'tiles69', 'wide12',
                                  'quad', 'line', 'cubic'])
def test_resident_lean_owner_bit_exact(eng, case):
    """DEMC, thinning 1, no trace, no shared parameters: the owner of a chain runs
    lean_resident_owner instead of the general run_chain -- same rows, counters and best fits
    as the oracle (and so as the general chain, which the traced tests above run)."""
    orc, gpu = run_pair(eng, _lean_case(case), trace=False)
    plan = gpu.plan
    assert plan['resident'] == 1 and plan['fast'] == 4 and plan['recoveries'] == 0, plan
    assert_same(orc, gpu, trace=False)
    if case == 'tight':
        assert gpu.state()['outbounds'].sum() > 50
    gpu.close()


@pytest.mark.parametrize('kind', ['linear', 'poly'])
def test_resident_lean_owner_resume_abort_and_switch(eng, kind):
    prob = linear(nmodel=9, nchains=48, niter=40) if kind == 'linear' else poly(3, niter=40)
    with env(MC3HIP_PERSIST_REFS=7):
        gpu = eng.DeviceChains(**prob)
        assert gpu.plan['resident'] == 1 and gpu.plan['persistent'] == 7
        assert gpu.plan['fast'] == 4
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nrefresh=40, trace=False, nthreads=4)
        for n in (3, 23, 14):
            gpu.run(n)
        assert_same(orc, gpu, trace=False)
        gpu.close()
    with env(MC3HIP_TEST_ABORT_AT=9):
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['recoveries'] == 1
        assert_same(orc, gpu, trace=False)
        gpu.close()
    with env(MC3HIP_NO_RS_LEAN=1):
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['resident'] == 1 and gpu.plan['fast'] == 0
        assert_same(orc, gpu, trace=False)
        gpu.close()


# ---- the owners as waves of their own, the blocks of chains made up at run time ----------
# (resident_async_kernel, resident.h: DEMC, thinning 1, no trace, no shared parameters, design
# matrices through the matrix cores, at most three chains to own per workgroup)
def _async_case(case):
    if case == 'c5like':        # three blocks of operand rows (the third in LDS), priors
        return linear(ndata=30000, nmodel=10, nchains=96, niter=30)
    if case == 'one_block':     # 2..3 coefficients + the data: one block of operand rows
        return linear(ndata=20000, nmodel=3, nchains=40, niter=30)
    if case == 'two_blocks':
        return linear(ndata=20000, nmodel=7, nchains=64, niter=30)
    if case == 'four_blocks':   # 12..15 coefficients: two blocks of operand rows in LDS
        return linear(ndata=20000, nmodel=14, nchains=48, niter=20)
    if case == 'ragged':        # columns of 32, 32 and 19 chains: a block of three behind one of 16
        return linear(ndata=24000, nmodel=9, nchains=83, niter=30)
    if case == 'epsilon':
        return linear(ndata=20000, nmodel=8, nchains=64, niter=30, fepsilon=0.01)
    if case == 'tight':         # most proposals out of bounds: sums consumed, not used
        prob = linear(nmodel=6, nchains=64, niter=60)
        prob['pmin'] = prob['params'] - 1.2*np.abs(prob['pstep'])
        prob['pmax'] = prob['params'] + 1.2*np.abs(prob['pstep'])
        return prob
    if case == 'tiles69':       # the owner collects the sums 64 at a time
        return linear(ndata=140000, nmodel=10, nchains=96, niter=6)
    if case == 'long_column':   # 128 chains per column: both halves of the scheduler's lanes
        return linear(ndata=100000, nmodel=10, nchains=512, niter=6)
    raise ValueError(case)


@pytest.mark.parametrize('case', ['c5like', 'one_block', 'two_blocks', 'four_blocks', 'ragged',
                                  'epsilon', 'tight', 'tiles69', 'long_column'])
def test_resident_async_bit_exact(eng, case):
    """Owner waves apart from the tile waves, a scheduler wave per column making up the blocks
    of sixteen chains from the proposals that are there: which chains share a block, and
    when, changes no bit -- the same rows, counters and best fits as the oracle, and as the
    kernel that goes through an iteration in step (MC3HIP_NO_RS_ASYNC=1)."""
    prob = _async_case(case)
    # (MC3HIP_RS_ASYNC=1: whatever the size -- the library itself takes this kernel from six
    # blocks of sixteen chains per column on, decide_persist)
    kw = dict({'long_column': dict(MC3HIP_RESIDENT_MAXCOL=4),
               'ragged': dict(MC3HIP_RESIDENT_MAXCOL=3)}.get(case, {}), MC3HIP_RS_ASYNC=1)
    with env(**kw):
        orc, gpu = run_pair(eng, prob, trace=False)
        plan = gpu.plan
        assert plan['resident'] == 1 and plan['fast'] == 4 and plan['recoveries'] == 0, plan
        assert plan['owner_waves'] >= 1, plan
        assert_same(orc, gpu, trace=False)
        if case == 'tight':
            assert gpu.state()['outbounds'].sum() > 50
        gpu.close()
        with env(MC3HIP_NO_RS_ASYNC=1):
            orc, gpu = run_pair(eng, prob, trace=False)
            assert gpu.plan['resident'] == 1 and gpu.plan['owner_waves'] == 0, gpu.plan
            assert_same(orc, gpu, trace=False)
            gpu.close()


def test_resident_async_many_launches_resume_and_abort(eng):
    """Launches of 5 refreshes (the scheduler's rows are told apart by the launch's number),
    the run split over several calls; a wave that gives up at refresh 9 ends the launch for
    everybody -- tile waves, owners, schedulers -- and the run is redone from the checkpoint."""
    prob = linear(ndata=24000, nmodel=10, nchains=80, niter=60)
    with env(MC3HIP_PERSIST_REFS=5, MC3HIP_RS_ASYNC=1):
        gpu = eng.DeviceChains(**prob)
        assert gpu.plan['owner_waves'] >= 1 and gpu.plan['persistent'] == 5, gpu.plan
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nrefresh=60, trace=False, nthreads=4)
        for n in (1, 4, 22, 33):
            gpu.run(n)
        assert_same(orc, gpu, trace=False)
        gpu.close()
    with env(MC3HIP_TEST_ABORT_AT=9, MC3HIP_RS_ASYNC=1):
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['recoveries'] == 1
        assert_same(orc, gpu, trace=False)
        gpu.close()


def test_resident_async_thinning_fills_a_launch(eng):
    """A launch of 40 refreshes at thinning 3: 120 rounds of proposals -- the scheduler's rows
    are sized for the rounds of a launch, not for its refreshes."""
    prob = linear(ndata=20000, nmodel=7, nchains=40, niter=120, sampler='snooker', thinning=3)
    with env(MC3HIP_PERSIST_REFS=40, MC3HIP_RS_ASYNC=1):
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['owner_waves'] >= 1 and gpu.plan['persistent'] == 40, gpu.plan
        assert gpu.plan['recoveries'] == 0
        assert_same(orc, gpu, trace=False)
        gpu.close()


def test_resident_async_shapes_it_leaves_to_the_kernel_in_step(eng):
    """More than three chains to own per workgroup (10 tiles x 1 column, 40 chains), four blocks
    of operand rows beside the general chain as owner: the kernel in step, as before."""
    prob = linear(ndata=20000, nmodel=7, nchains=40, niter=12)
    with env(MC3HIP_RESIDENT_MAXCOL=1, MC3HIP_RS_ASYNC=1):
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['resident'] == 1 and gpu.plan['owner_waves'] == 0, gpu.plan
        assert_same(orc, gpu, trace=False)
        gpu.close()
    with env(MC3HIP_RS_ASYNC=1):
        orc, gpu = run_pair(eng, linear(nmodel=14, nchains=40, niter=10, sampler='snooker'))
        assert gpu.plan['resident'] == 1 and gpu.plan['owner_waves'] == 0, gpu.plan
        assert_same(orc, gpu)
        gpu.close()
    # small problems: fewer than six (DEMC) / four (the other samplers) blocks of sixteen chains
    # per column -- an iteration in step is shorter than a chain's round through the hand-offs
    for smp in ('demc', 'snooker'):
        orc, gpu = run_pair(eng, linear(ndata=30000, nmodel=10, nchains=96, niter=6,
                                        sampler=smp), trace=False)
        assert gpu.plan['resident'] == 1 and gpu.plan['owner_waves'] == 0, gpu.plan
        assert_same(orc, gpu, trace=False)
        gpu.close()


@pytest.mark.parametrize('kw', [dict(sampler='snooker'), dict(sampler='mrw'),
                                dict(sampler='demc', thinning=3),
                                dict(sampler='snooker', thinning=2, nmodel=10),
                                dict(sampler='snooker', fepsilon=0.01, nmodel=11),
                                dict(sampler='mrw', nmodel=3), dict(sampler='demc', shared=True),
                                dict(sampler='snooker', nmodel=10, nchains=96, ndata=40000)])
def test_resident_async_general_owner_bit_exact(eng, kw):
    """The other samplers, thinning, shared parameters, the trace: the general chain as owner
    wave of resident_async_kernel (what it calls at the tile phase only turns the proposal
    rows' parity over) -- against the oracle, with the per-iteration trace, and against the
    kernel in step."""
    kw = dict(kw)
    shared = kw.pop('shared', False)
    prob = linear(**dict(dict(nmodel=7, nchains=40, niter=20), **kw))
    if shared:
        prob['pstep'] = prob['pstep'].copy()
        prob['params'] = prob['params'].copy()
        prob['pstep'][5] = -2
        prob['params'][5] = prob['params'][1]
    with env(MC3HIP_RS_ASYNC=1):
        orc, gpu = run_pair(eng, prob)
        assert gpu.plan['resident'] == 1 and gpu.plan['owner_waves'] >= 1, gpu.plan
        assert gpu.plan['fast'] == 0 and gpu.plan['recoveries'] == 0, gpu.plan
        assert_same(orc, gpu)
        gpu.close()
    with env(MC3HIP_NO_RS_ASYNC=1):
        orc, gpu = run_pair(eng, prob)
        assert gpu.plan['owner_waves'] == 0
        assert_same(orc, gpu)
        gpu.close()
    with env(MC3HIP_TEST_ABORT_AT=7, MC3HIP_RS_ASYNC=1):
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['recoveries'] == 1
        assert_same(orc, gpu, trace=False)
        gpu.close()


@pytest.mark.parametrize('sampler,slow', [('mrw', '0:400'), ('snooker', '1:400'),
                                          ('demc', '2:300'), ('mrw', '2:1500')])
def test_resident_async_columns_drift_apart(eng, sampler, slow):
    """One column's tile waves held back before every block (test hook MC3HIP_TEST_SLOW_COLUMN =
    column:ticks of the 100 MHz clock): with the random walk and the snooker update no chain
    waits for another, so the other columns run ahead as far as the hand-off lets them -- and
    the columns' chains change from round to round (112 chains = 7 blocks for 4 columns: three
    of them carry a block more, in turns), so a chain enters a column that is rounds ahead of the
    one it leaves.  A proposal row is taken for a round only with its owner's word that it IS
    that round's (DevProblem::rs_tag), and a wave enters a round only when its workgroup has
    left the slots of the round before last: 300 rounds, every row as the oracle's."""
    prob = linear(ndata=20000, nmodel=7, nchains=112, niter=300, sampler=sampler)
    with env(MC3HIP_RS_ASYNC=1, MC3HIP_RESIDENT_MAXCOL=4, MC3HIP_TEST_SLOW_COLUMN=slow):
        orc, gpu = run_pair(eng, prob, trace=False)
        plan = gpu.plan
        assert plan['resident'] == 1 and plan['owner_waves'] >= 1 and plan['recoveries'] == 0, plan
        assert 'resident_async_kernel' in gpu.last_kernel
        assert_same(orc, gpu, trace=False)
        gpu.close()
