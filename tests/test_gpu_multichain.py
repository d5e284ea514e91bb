"""More chains than workgroups can be co-resident: a workgroup of the persistent kernels
advances SEVERAL chains per refresh (run_chain MC, mc3_amd/csrc/run_chain.h), as the
reference's workers loop over their ncpp chains (mc3/chain.py:163,190-191).  Results must
not depend on it: everything is compared with the oracle for exact equality, the launch must
stay persistent (`plan.persistent > 0`, `chains_per_workgroup > 1`) and must not have been
redone (`recoveries == 0`).

Big ensembles take the path by themselves (4096 / 8192 chains x ndata 1000: twice / four
times what fits; 512 / 1024 chains x 2^14 wavelet points: one 8-wave workgroup per CU);
MC3HIP_PERSIST_CAPACITY (a test hook: pretend that fewer workgroups fit) sends small
ensembles of every kernel flavour down the same path."""
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


def check_plan(gpu, turns=None, fast=None):
    plan = gpu.plan
    assert plan['persistent'] > 0 and plan['recoveries'] == 0, plan
    assert plan['chains_per_workgroup'] > 1, plan
    if turns is not None:
        assert plan['chains_per_workgroup'] == turns, plan
    if fast is not None:
        assert plan['fast'] == fast, plan
    return plan


@pytest.mark.parametrize('nchains,sampler,nref', [
    (4096, 'demc', 30), (4096, 'snooker', 24), (8192, 'demc', 16), (8192, 'snooker', 12),
    (2049, 'demc', 40), (5000, 'mrw', 20)])
def test_big_ensembles_stay_on_the_persistent_path(eng, nchains, sampler, nref):
    """C2-shaped (quadratic, ndata 1000) beyond the 2048 co-resident single-wave workgroups."""
    prob = H.quad_problem(ndata=1000, nchains=nchains, nsamples=nchains*nref, sampler=sampler)
    orc, gpu = run_pair(eng, prob, trace=False)
    # (two to four chains per wave: the hand-laid-out loops, lean_demc_mc_kernel,
    # lean_snooker_mc_kernel, lean_mrw_mc_kernel; with three or four chains per wave those two
    # fit one wave per SIMD, so 5000 / 8192 chains of theirs take more turns on the general lean
    # instance)
    plan = check_plan(gpu)
    assert plan['fast'] == (2 if plan['chains_per_workgroup'] <= 4 else 1), plan
    if sampler == 'demc' or nchains <= 4096:
        assert plan['fast'] == 2, plan
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('nchains,nref', [(512, 6), (1024, 4), (300, 8)])
def test_wavelet_ensembles_beyond_one_workgroup_per_cu(eng, nchains, nref):
    """C4-shaped (2^14 wavelet points, 8 waves and ~155 KB of LDS per chain: one workgroup per
    CU) with more chains than CUs."""
    prob = H.wavelet_problem(ndata=16384, nchains=nchains, nsamples=nchains*nref)
    orc, gpu = run_pair(eng, prob, trace=False)
    check_plan(gpu, fast=3)                  # (nine waves, the chain wave takes the turns)
    assert_same(orc, gpu, trace=False)
    gpu.close()


def forced(monkeypatch, capacity):
    monkeypatch.setenv('MC3HIP_PERSIST_CAPACITY', str(capacity))


@pytest.mark.parametrize('sampler,thinning', [('demc', 1), ('demc', 3), ('snooker', 1),
                                              ('snooker', 2), ('mrw', 1), ('mrw', 2)])
@pytest.mark.parametrize('nchains,capacity', [(16, 5), (23, 4), (64, 1)])
def test_lean_instances_with_several_chains_per_wave(eng, monkeypatch, sampler, thinning,
                                                     nchains, capacity):
    """The lean one-wave kernels (persist_fast_kernel<.., MC>): every sampler, thinning, uneven
    splits (23 chains over 4 workgroups: 6, 6, 6, 5), ONE workgroup for the whole ensemble."""
    forced(monkeypatch, capacity)
    prob = H.quad_problem(ndata=300, nchains=nchains, nsamples=nchains*40*thinning,
                          sampler=sampler, thinning=thinning)
    orc, gpu = run_pair(eng, prob, trace=False)
    turns = -(-nchains//capacity)
    lean_loop = thinning == 1 and turns <= 4
    check_plan(gpu, turns=turns, fast=2 if lean_loop else 1)
    assert_same(orc, gpu, trace=False)
    gpu.close()
    if lean_loop:
        # the general chain behind the switch (several chains per workgroup as well)
        monkeypatch.setenv('MC3HIP_NO_LEAN', '1')
        orc, gpu = run_pair(eng, prob, trace=False)
        check_plan(gpu, turns=turns, fast=0)
        assert_same(orc, gpu, trace=False)
        gpu.close()


@pytest.mark.parametrize('sampler', ['demc', 'snooker', 'mrw'])
def test_lean_fixed_coefficient_and_streamed_tiles(eng, monkeypatch, sampler):
    forced(monkeypatch, 3)
    prob = H.quad_problem(ndata=500, nchains=10, nsamples=10*60, sampler=sampler)
    prob['pstep'] = np.array([1.0, 0.0, 0.1])           # FX: the linear coefficient is fixed
    orc, gpu = run_pair(eng, prob, trace=False)
    check_plan(gpu, turns=4, fast=2)
    assert_same(orc, gpu, trace=False)
    gpu.close()
    prob = H.quad_problem(ndata=3000, nchains=10, nsamples=10*40, sampler=sampler)   # MULTI
    orc, gpu = run_pair(eng, prob, trace=False)
    # (four chains per wave and two register-resident tiles: the general instance)
    check_plan(gpu, turns=4, fast=1)
    assert_same(orc, gpu, trace=False)
    gpu.close()
    forced(monkeypatch, 4)
    orc, gpu = run_pair(eng, prob, trace=False)
    check_plan(gpu, turns=3, fast=2 if sampler == 'demc' else 1)
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('sampler,thinning,eps', [('snooker', 1, 0.0), ('demc', 2, 0.01),
                                                  ('mrw', 1, 0.0), ('snooker', 3, 0.02)])
def test_general_kernel_with_several_chains_per_workgroup(eng, monkeypatch, sampler, thinning,
                                                          eps):
    """persist_kernel<.., MC>: shared and fixed parameters, priors, tight bounds, epsilon, the
    per-step trace (chisq and flags of every iteration of every chain)."""
    forced(monkeypatch, 3)
    prob = H.shared_problem(nchains=8, nsamples=8*50*thinning, sampler=sampler,
                            thinning=thinning, fepsilon=eps)
    orc, gpu = run_pair(eng, prob, trace=True)
    check_plan(gpu, turns=3, fast=0)
    assert_same(orc, gpu, trace=True)
    gpu.close()


@pytest.mark.parametrize('wpc', ['4'])
def test_general_kernel_wide_workgroups(eng, monkeypatch, wpc):
    forced(monkeypatch, 2)
    monkeypatch.setenv('MC3HIP_WPC', wpc)
    prob = H.quad_problem(ndata=6000, nchains=7, nsamples=7*30, sampler='snooker', fepsilon=0.01)
    orc, gpu = run_pair(eng, prob, trace=True)
    plan = check_plan(gpu, turns=4, fast=0)
    assert plan['waves_per_chain'] == int(wpc)
    assert_same(orc, gpu, trace=True)
    gpu.close()


@pytest.mark.parametrize('ndata,sampler', [(64, 'snooker'), (2048, 'demc'), (4096, 'snooker'),
                                           (8192, 'mrw')])
def test_wavelet_kernels_with_several_chains_per_workgroup(eng, monkeypatch, ndata, sampler):
    forced(monkeypatch, 3)
    prob = H.wavelet_problem(ndata=ndata, nchains=8, nsamples=8*25, sampler=sampler)
    # (8192 points: only the nine-wave kernel with its hand-structured chain wave takes turns --
    # untraced; the fused eight-wave kernel that served the traced run went with round 4, such
    # a run is one launch per refresh now)
    trace = ndata < 8192
    orc, gpu = run_pair(eng, prob, trace=trace)
    check_plan(gpu, turns=3, fast=0 if trace else 3)
    assert_same(orc, gpu, trace=trace)
    gpu.close()
    if not trace:
        orc, gpu = run_pair(eng, prob, trace=True)
        assert gpu.plan['persistent'] == 0
        assert_same(orc, gpu, trace=True)
        gpu.close()


def test_linear_model_and_resumed_launches(eng, monkeypatch):
    """A design-matrix model below the tiled plan; the run cut into pieces (every launch
    reloads the waiting states from the rows of the refresh before)."""
    forced(monkeypatch, 4)
    prob = H.linear_problem(ndata=700, nmodel=6, nchains=12, nsamples=12*60)
    gpu = eng.DeviceChains(**prob)
    orc = H.make_oracle(prob, plan=gpu.plan)
    Z0, lp0 = H.initial_population(orc)
    orc.set_population(Z0, lp0)
    gpu.set_population(Z0, lp0)
    orc.run(nthreads=4)
    for n in (1, 7, 20, 32):
        gpu.run(n)
    check_plan(gpu, turns=3, fast=0)
    assert_same(orc, gpu, trace=False)
    gpu.close()


def test_a_launch_that_gives_up_is_redone(eng, monkeypatch):
    """The test hook makes chain 0 give up at refresh 9: the run is redone from the checkpoint
    with one launch per refresh, identical to the oracle."""
    forced(monkeypatch, 3)
    monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '9')
    prob = H.quad_problem(ndata=300, nchains=10, nsamples=10*30, sampler='snooker')
    orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert plan['recoveries'] == 1 and plan['persistent'] == 0, plan
    assert_same(orc, gpu, trace=False)
    gpu.close()


def test_switch_off(eng, monkeypatch):
    monkeypatch.setenv('MC3HIP_NO_MULTICHAIN', '1')
    forced(monkeypatch, 3)
    prob = H.quad_problem(ndata=300, nchains=10, nsamples=10*20, sampler='demc')
    orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    assert plan['persistent'] == 0 and plan['chains_per_workgroup'] == 1, plan
    assert_same(orc, gpu, trace=False)
    gpu.close()


def layout_switch(monkeypatch, layout, case):
    """'rows': the chains of a wave side by side (kernels_rows.h).  (Until round 6 there was a
    second layout behind a switch -- one after another on the lean instance of run_chain -- for
    the shapes the rows loops serve; those instances went with the switch.)"""


@pytest.mark.parametrize('layout', ['rows'])
@pytest.mark.parametrize('turns', [2, 3, 4])
@pytest.mark.parametrize('case', ['resume', 'abort', 'tight', 'priors', 'ragged'])
def test_lean_demc_loop_with_several_chains_per_wave(eng, monkeypatch, case, turns, layout):
    """lean_demc_rows_kernel<.., T> (and run_chain MC behind the switch): launches of a few refreshes (each
    starts a batch of draws anew and reloads the chains' states), a give-up, bounds that reject
    most proposals, Gaussian priors (FX), an ensemble that leaves the last turn of some waves
    empty."""
    layout_switch(monkeypatch, layout, case)
    nchains = {2: 10, 3: 14, 4: 23}[turns] if case == 'ragged' else 5*turns
    forced(monkeypatch, 5 if case != 'ragged' else -(-nchains//turns))
    prob = H.quad_problem(ndata=400, nchains=nchains, nsamples=nchains*70, sampler='demc')
    if case == 'tight':
        prob.update(params=np.array([4.0, -2.5, 0.5]), pmin=np.array([3.5, -2.8, 0.45]),
                    pmax=np.array([6.5, -2.2, 0.56]))
    if case == 'priors':
        prob.update(prior=np.array([4.0, 0.0, 0.0]), priorlow=np.array([0.5, 0.0, 0.0]),
                    priorup=np.array([0.7, 0.0, 0.0]))
    if case == 'abort':
        monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '9')
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['recoveries'] == 1
        assert_same(orc, gpu, trace=False)
        gpu.close()
        return
    if case == 'resume':
        monkeypatch.setenv('MC3HIP_PERSIST_REFS', '7')
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 30, 37):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    check_plan(gpu, turns=turns, fast=2 if layout == 'rows' else 1)
    assert_same(orc, gpu, trace=False)
    if case == 'tight':
        assert gpu.state()['outbounds'].sum() > 50
    gpu.close()


@pytest.mark.parametrize('layout', ['rows'])
@pytest.mark.parametrize('turns', [2, 3, 4])
@pytest.mark.parametrize('sampler', ['snooker', 'mrw'])
@pytest.mark.parametrize('case', ['resume', 'abort', 'tight', 'priors', 'ragged', 'thin'])
def test_lean_snooker_and_mrw_loops_with_several_chains_per_wave(eng, monkeypatch, case, sampler,
                                                                 turns, layout):
    """lean_snooker_rows_kernel / lean_mrw_rows_kernel<.., T> (and run_chain MC behind the switch):
    launches of a few refreshes, a
    give-up, bounds that reject most proposals, Gaussian priors (FX), an ensemble that leaves
    the last turn of some waves empty, a history of one row per chain (snooker draws from the
    newest rows: the wait for a row that is still the sentinel)."""
    layout_switch(monkeypatch, layout, case)
    nchains = {2: 10, 3: 14, 4: 23}[turns] if case == 'ragged' else 5*turns
    forced(monkeypatch, 5 if case != 'ragged' else -(-nchains//turns))
    prob = H.quad_problem(ndata=400, nchains=nchains, nsamples=nchains*70, sampler=sampler)
    if case == 'tight':
        prob.update(params=np.array([4.0, -2.5, 0.5]), pmin=np.array([3.5, -2.8, 0.45]),
                    pmax=np.array([6.5, -2.2, 0.56]))
    if case == 'priors':
        prob.update(prior=np.array([4.0, 0.0, 0.0]), priorlow=np.array([0.5, 0.0, 0.0]),
                    priorup=np.array([0.7, 0.0, 0.0]))
    if case == 'thin':
        prob.update(hsize=1)
    if case == 'abort':
        monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '9')
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['recoveries'] == 1
        assert_same(orc, gpu, trace=False)
        gpu.close()
        return
    if case == 'resume':
        monkeypatch.setenv('MC3HIP_PERSIST_REFS', '7')
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 30, 37):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    check_plan(gpu, turns=turns, fast=2 if layout == 'rows' else 1)
    assert_same(orc, gpu, trace=False)
    if case == 'tight':
        assert gpu.state()['outbounds'].sum() > 50
    gpu.close()
    # the general chain behind the switch
    monkeypatch.setenv('MC3HIP_NO_LEAN', '1')
    if case != 'resume':
        orc, gpu = run_pair(eng, prob, trace=False)
        check_plan(gpu, turns=turns, fast=0)
        assert_same(orc, gpu, trace=False)
        gpu.close()


@pytest.mark.parametrize('case', ['plain', 'tight', 'wide', 'abort', 'resume', 'nospec', 'noone'])
@pytest.mark.parametrize('ndata,sampler', [(8192, 'snooker'), (16384, 'mrw'), (16384, 'snooker')])
def test_wavelet_one_chain_per_workgroup_speculates_on_reject(eng, monkeypatch, ndata, sampler, case):
    """One chain per workgroup on the pipelined kernel: the next proposal is issued while the one
    in flight is undecided, as if that one were rejected; after an accept it is squashed and issued
    again from the new state.  Nothing of a squashed proposal may show: same rows, same counters
    (accepts, out-of-bounds), same best fit as the checker -- with bounds that reject a third of
    the proposals before the likelihood, with none, across launches, through a give-up, and behind
    the switches (the nine-wave kernel: one proposal at a time; the general chain wave of the
    pipeline).  150 refreshes: the batch of draws (21 turns) is refilled seven times, squashes
    rewind across refills."""
    prob = H.wavelet_problem(ndata=ndata, nchains=6, nsamples=6*150, sampler=sampler)
    if case == 'tight':
        prob.update(pmin=np.array([2.6, -2.6, 0.47, 0.0, 0.3, 0.8]),
                    pmax=np.array([3.4, -2.2, 0.53, 2.0, 0.7, 1.2]))
    if case == 'wide':
        prob.update(pmin=np.full(6, -1e3), pmax=np.full(6, 1e3))
    if case == 'nospec':
        monkeypatch.setenv('MC3HIP_NO_CW_SPEC', '1')
    if case == 'noone':
        # (snooker: the chain wave that serves any number of chains per workgroup instead of
        # the one with its state in registers and its draws a batch ahead, pipe_chain_wave_one
        # -- which every other snooker case here runs)
        monkeypatch.setenv('MC3HIP_NO_CW_ONE', '1')
    if case == 'abort':
        monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '40')
    if case == 'resume':
        monkeypatch.setenv('MC3HIP_PERSIST_REFS', '11')
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 50, 97):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    plan = gpu.plan
    if case == 'abort':                  # (the launch gave up: redone one launch per refresh)
        assert plan['recoveries'] == 1, plan
    else:
        assert plan['persistent'] > 0 and plan['fast'] == 3 and plan['recoveries'] == 0, plan
        assert plan['chains_per_workgroup'] == 1, plan
    assert_same(orc, gpu, trace=False)
    st = gpu.state()
    assert st['numaccept'].sum() > 20               # (accepts there were: squashes too)
    if case == 'tight':
        assert st['outbounds'].sum() > 50
    gpu.close()


@pytest.mark.parametrize('nchains,capacity,turns', [(5, 4, 2), (9, 2, 5), (7, 1, 7)])
@pytest.mark.parametrize('sampler', ['snooker', 'mrw'])
def test_wavelet_pipeline_uneven_and_deep(eng, monkeypatch, sampler, nchains, capacity, turns):
    """The pipelined kernel where its slots do not fill by themselves: a workgroup with ONE chain
    (5 chains over 3 workgroups: 2, 2, 1 -- every other slot is a bubble, the chain's previous turn
    must be decided first), five and four chains, and the whole ensemble in one workgroup (every
    history row snooker asks for is this workgroup's own: what is still in flight is waited for
    with an empty slot, never with a spin)."""
    forced(monkeypatch, capacity)
    prob = H.wavelet_problem(ndata=8192, nchains=nchains, nsamples=nchains*80, sampler=sampler)
    orc, gpu = run_pair(eng, prob, trace=False)
    check_plan(gpu, turns=turns, fast=3)
    assert_same(orc, gpu, trace=False)
    gpu.close()


@pytest.mark.parametrize('case', ['plain', 'epsilon', 'tight', 'resume', 'abort', 'general'])
@pytest.mark.parametrize('ndata,sampler', [(8192, 'snooker'), (16384, 'demc'), (8192, 'mrw'),
                                           (16384, 'snooker')])
def test_wavelet_chain_wave_takes_turns(eng, monkeypatch, ndata, sampler, case):
    """The wavelet kernels with a wave of their own for the chain and several chains per
    workgroup.  snooker and mrw: eleven waves, the chains pipelined (wavelet_pipe_kernel: the tail
    of one chain's likelihood, the noise terms and the chain itself on three helper waves while
    the eight compute waves are at the next chain's stage A); DEMC: nine waves, the chain wave
    advances the chains in turn and prepares, during a likelihood, the NEXT turn's proposal.
    11 chains over 4 workgroups: 3, 3, 3, 2."""
    forced(monkeypatch, 4)
    prob = H.wavelet_problem(ndata=ndata, nchains=11, nsamples=11*50, sampler=sampler)
    if case == 'epsilon':
        prob.update(fepsilon=0.03, fgamma=0.8)
    if case == 'tight':
        prob.update(pmin=np.array([2.6, -2.6, 0.47, 0.0, 0.3, 0.8]),
                    pmax=np.array([3.4, -2.2, 0.53, 2.0, 0.7, 1.2]))
    if case == 'abort':
        monkeypatch.setenv('MC3HIP_TEST_ABORT_AT', '9')
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['recoveries'] == 1
        assert_same(orc, gpu, trace=False)
        gpu.close()
        return
    if case == 'general':
        # the general chain behind the switch: nothing but the hand-structured chain wave takes
        # turns on this geometry (the fused eight-wave kernel went with round 4), so the run
        # is one launch per refresh -- the same results
        monkeypatch.setenv('MC3HIP_NO_CW_LEAN', '1')
        orc, gpu = run_pair(eng, prob, trace=False)
        assert gpu.plan['persistent'] == 0 and gpu.plan['fast'] == 0, gpu.plan
        assert_same(orc, gpu, trace=False)
        gpu.close()
        return
    if case == 'resume':
        monkeypatch.setenv('MC3HIP_PERSIST_REFS', '7')
        gpu = eng.DeviceChains(**prob)
        orc = H.make_oracle(prob, plan=gpu.plan)
        Z0, lp0 = H.initial_population(orc)
        orc.set_population(Z0, lp0)
        gpu.set_population(Z0, lp0)
        orc.run(nthreads=4)
        for n in (1, 2, 30, 17):
            gpu.run(n)
    else:
        orc, gpu = run_pair(eng, prob, trace=False)
    check_plan(gpu, turns=3, fast=3)
    assert_same(orc, gpu, trace=False)
    gpu.close()
