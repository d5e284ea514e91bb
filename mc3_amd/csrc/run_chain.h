// run_chain.h -- the chain update: one device function, instantiated as every chain kernel.
#pragma once
#include "core.h"
#include "chisq.h"
#include "handoff.h"

namespace mc3hip {

struct NoHook {};

// One chain, refreshes [R0, R0 + nref).  PERSIST = false: one refresh per launch,
// other chains are seen through rows a previous launch wrote (plain loads).
// PERSIST = true: many refreshes per launch; the tile, the chain state and all
// constants stay in registers, and the other chains' rows arrive through the
// in-launch hand-off above.
//
// SMP >= 0 fixes the sampler at compile time.  FAST is the lean instance for the
// commonest shape -- one wave per chain, Gaussian chisq, every parameter a free
// polynomial coefficient, no priors / shared parameters / support distribution /
// trace: the proposal goes from lanes to the model through v_readlane instead of
// LDS and all the dead generality (and its scalar-register pressure) is compiled
// out.  Same arithmetic, same results.
//
// PHASE splits one iteration over three launches for the tiled plan: PH_PROPOSE
// stops after the proposal (written to global scratch), a tile kernel computes the
// data sums for all chains, PH_FINISH resumes at the Metropolis test.  jb..je is the
// range of iterations of the refresh this call covers (fused: 0..thinning).
// P2P (persistent only): the ensemble is sharded over GPUs; a chain publishes its new
// row into EVERY rank's replica of Z (peer-to-peer stores over xGMI, system scope) and
// polls its local replica, so the in-launch hand-off spans GPUs and no collective or
// kernel boundary separates iterations.
// TH = 1: exactly one iteration per refresh (thinning = 1, the default), known at
// compile time so that the inner loop and its control flow disappear.
// PH_RESIDENT (resident tiled plan, see ResidentTile): this WAVE owns chain `own_chain` of a
// workgroup whose waves all take part in the tile phase of every iteration; the data sums
// come back through `hook`; the wave leaves the loop only when the whole workgroup does.
// PH_CHAINWAVE (wavelet likelihood, kernels_wavelet.h): this wave does nothing but run the
// chain; the other waves of the workgroup compute the likelihood of every proposal it
// publishes.  As PH_RESIDENT, but the wave's registers are its own: snooker rows are asked
// for an iteration ahead.
// MC (persistent only): the ensemble has more chains than workgroups can be co-resident, so a
// workgroup owns SEVERAL chains -- local chains b, b + G, b + 2 G, ... of a grid of G
// workgroups, P.mc_chains at most -- and advances them one after another inside every
// refresh (the reference's workers do the same with their ncpp chains, mc3/chain.py:163,
// 190-191).  What a chain carries from one refresh to the next (its state, chisq, best
// chisq, accept count) waits in LDS between its turns; the data tile, the bounds and every
// other chain-invariant stay in registers.  The hand-off protocol is unchanged -- a chain
// waits only for rows of the PREVIOUS refresh, and every workgroup finishes refresh R - 1
// for all its chains before it starts R -- so co-residency of the G workgroups is all that
// is needed; the draws that are computed one iteration ahead are those of the NEXT TURN
// (next chain of this workgroup, or its first chain in the next refresh).
template <int WPC, int MT, bool WL, bool MULTI, bool PERSIST, int SMP = -1, bool FAST = false,
          int PHASE = PH_FUSED, bool P2P = false, int TH = 0, bool FX = false, class H = NoHook,
          bool MC = false>
__device__ __forceinline__ void run_chain(const DevProblem &P, const long long R0,
                                          const int nref, const int jb = 0, int je = -1,
                                          H *hook = nullptr, const int own_chain = 0,
                                          double *own_smem = nullptr)
{
    if (je < 0)
        je = P.thinning;
    static_assert(!phase_hooked(PHASE) || (WPC == 1 && PERSIST && !FAST && !WL),
                  "the resident tiled plan runs the general one-wave persistent chain");
    static_assert(!FAST || (WPC == 1 && !WL && SMP >= 0 && MT >= 1 && MT <= 8),
                  "FAST needs one wave, Gaussian chisq, a compile-time sampler and degree");
    static_assert(!MC || (PERSIST && PHASE == PH_FUSED && !P2P),
                  "several chains per workgroup: the one-GPU persistent kernels");
    const int sampler = (SMP >= 0) ? SMP : P.sampler;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SmemLayout L = carve(phase_hooked(PHASE) ? own_smem : smem, P, 64 * WPC);
    MC3_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const bool wave0 = (WPC == 1) ? true : (tid < 64);
    int lc = (phase_hooked(PHASE)) ? own_chain : blockIdx.x;   // local chain (MC: of the turn)
    int c = P.chain0 + lc;                     // global chain id
    const int nf = P.nfree;
    const int zs = P.zs;                       // row stride of Z / snap0 (padded lines)
    const bool mine = wave0 && lane < nf;      // lane k of wave 0 owns free parameter k

    // Everything that does not depend on the proposal is requested first, so the
    // memory latencies overlap each other and the RNG arithmetic: the data tile
    // (registers, kept for all iterations of the launch), the chain state, the
    // per-parameter bounds and the fixed parameter values.
    DataTile<WPC, MT> tile;
    if constexpr (!WL && PHASE == PH_FUSED)
        tile.load(P);
    if constexpr (WL) {                        // wavelet: the data (or the abscissae) into LDS
        wavelet_stage<WPC, MT>(P, L.xs);
        block_sync<WPC>();
    }
    DataTile<WPC, MT> tile1;                   // lean + MULTI: points 1024..2047 resident too
    if constexpr (FAST && MULTI)
        tile1.load(P, VLANES);
    // own state = this chain's row of the first snapshot (same pages as the partner
    // rows requested a few hundred cycles later, so their translation is warm); in
    // the middle of a refresh (tiled plan, thinning > 1) it is the saved state
    const double *snap_first = (R0 == 0) ? P.snap0
                                         : P.Z + (size_t)(P.M0 + (R0 - 1) * P.nchains) * zs;
    // What refresh R sees of the OTHER chains: the rows of refresh R - 1 - zlag (DEMC partners)
    // and the history up to and including that refresh (snooker).  zlag = 0 is the strict
    // lockstep; the declared cadence of a run sharded over GPUs looks zlag refreshes further
    // back (mc3hip_problem.zlag), the initial snapshot standing in until there is such a refresh.
    const long long zlag = P.zlag;
    auto snapshot_of = [&](long long R) -> const double * {
        const long long Rs = R - 1 - zlag;
        return (Rs < 0) ? P.snap0 : P.Z + (size_t)(P.M0 + Rs * P.nchains) * zs;
    };
    auto history_of = [&](long long R) -> long long {
        const long long Rs = R - zlag;
        return P.M0 + (Rs > 0 ? Rs : 0) * P.nchains;
    };
    double xk = 0.0;
    if (mine)
        xk = (jb == 0) ? snap_first[(size_t)c * zs + lane] : P.X[(size_t)lc * nf + lane];
    double cur = P.chisq[lc];
    double best = P.best_chisq[lc];
    const long long nacc0 = MC ? 0 : P.numaccept[lc];
    // P2P: the ranks' replicas, their pointers read once (handoff.h PeerTable)
    PeerTable<double> peers_z, peers_lp;
    if constexpr (P2P) {
        peers_z.load(P.peerZ, P.npeers);
        peers_lp.load(P.peerLP, P.npeers);
    }
    (void)peers_z; (void)peers_lp;
    // MC: what the chains of this workgroup carry between their turns, in LDS behind the
    // areas of carve(): slot q = [state (zs doubles) | chisq | best chisq | accept count]
    const int mc_turns = MC ? P.mc_chains : 1;
    const int mc_stride = zs + 4;
    double *const mc_state = MC ? smem + P.mc_off : nullptr;
    if constexpr (MC) {
        if (wave0) {
            for (int q = 0; q < mc_turns; ++q) {
                const int l = (int)blockIdx.x + q * (int)gridDim.x;
                if (l >= P.nlocal)
                    break;
                double *slot = mc_state + q * mc_stride;
                if (mine)
                    slot[lane] = snap_first[(size_t)(P.chain0 + l) * zs + lane];
                if (lane == 0) {
                    slot[zs] = P.chisq[l];
                    slot[zs + 1] = P.best_chisq[l];
                    slot[zs + 2] = mc3_from_bits((unsigned long long)P.numaccept[l]);
                }
            }
        }
    }
    const int jfree = mine ? P.ifree[lane] : 0;
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    // (the lean instance has no support distribution, fepsilon = 0; its MRW flavour draws)
    const bool need_normal = FAST ? (SMP == MC3HIP_MRW)
                                  : ((sampler == MC3HIP_MRW) || (P.fepsilon != 0.0));
    const double pstep_k = (mine && need_normal) ? P.fstep[lane] : 0.0;
    const double fixed0 = (!FAST && wave0 && lane < P.nparams) ? P.params[lane] : 0.0;
    const double fixed1 = (!FAST && wave0 && lane + 64 < P.nparams) ? P.params[lane + 64] : 0.0;
    long long nacc = 0;
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    bool alive = true;
    constexpr int NKF = (FX && MT >= 1 && MT <= 8) ? MT : 1;
    int kf[NKF];
    double fx[NKF];
    PriorLanes<NKF> priors;
    if constexpr (FX) {
        coefficient_map<NKF>(P, kf, fx);
        priors.init(P, kf, fx, lane);
    }
    (void)kf; (void)fx; (void)priors;

    if constexpr (PERSIST && !phase_hooked(PHASE)) {     // (resident: the kernel has looked)
        // a launch queued behind one that gave up must not run: its inputs are missing
        if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(
                &ctl_rw[CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
            return;
    }
    MC3_STAMP(1);

    // The index / acceptance draws depend only on (seed, iteration, chain): they
    // are wave-uniform, so Philox runs on the scalar unit (every wave repeats it),
    // and the next iteration's are computed while this iteration's partner rows
    // are in flight.
    Draws nxt = make_draws(P, (TH == 1) ? (uint64_t)R0 : (uint64_t)(R0 * P.thinning + jb), c,
                           sampler);
    // Persistent snooker: the three Z rows of an iteration (chain.py:195-205) come from the
    // whole history, almost always rows published long ago, so they are requested one
    // iteration ahead -- as soon as the previous iteration's rows have been consumed -- and
    // the fabric round trip hides behind the chisq.  A row of the newest refresh may still
    // be the sentinel when it is looked at; the usual poll loop then asks again.
    const double *sa[3] = {nullptr, nullptr, nullptr};
    double sv[3] = {0.0, 0.0, 0.0};
    auto request_snooker = [&](const Draws &D, long long zsz) {
        const long long iR1 = (long long)bounded(D.d0.q0, (uint64_t)zsz);
        long long iR2 = 1 + (long long)bounded(D.d0.q1, (uint64_t)(zsz - 1));
        if (iR2 == iR1)
            iR2 = 0;
        const long long iz = (long long)bounded(D.d1.q1, (uint64_t)zsz);
        const int kk = mine ? lane : 0;
        sa[0] = P.Z + (size_t)iR1 * zs + kk;
        sa[1] = P.Z + (size_t)iR2 * zs + kk;
        sa[2] = P.Z + (size_t)iz * zs + kk;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            sv[q] = load_published<P2P>(sa[q]);
    };
    // (not in the resident tiled plan: rows and batch would sit in registers across the tile
    // phase, which needs them all; there the rows are asked for where they are used)
    constexpr bool SNOOKER_AHEAD =
        PERSIST && PHASE != PH_RESIDENT && (SMP < 0 || SMP == MC3HIP_SNOOKER);
    DrawBatch batch;
    batch.w0 = batch.w1 = batch.w2 = batch.w3 = 0u;
    batch.base = -(long long)DRAW_BATCH - 1;      // empty
    if constexpr (SNOOKER_AHEAD) {
        if (sampler == MC3HIP_SNOOKER && wave0)
            request_snooker(nxt, history_of(R0));
    }
    // diagnostic build only: per-phase tick sums kept in registers, stored once at exit
    unsigned long long tk_prev3 = 0, tk_prevend = 0;
    unsigned long long ta8 = 0, ta9 = 0, ta10 = 0, ta11 = 0, ta12 = 0, ta13 = 0, ta14 = 0, ta15 = 0;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0;
    (void)tr0; (void)tr1; (void)tr2; (void)tr3;
    (void)ta14; (void)ta15;
    (void)tk_prev3; (void)tk_prevend;
    (void)ta8; (void)ta9; (void)ta10; (void)ta11; (void)ta12; (void)ta13;
    bool give_up = false;                       // resident plan: test hook fired
    (void)give_up;
    auto normal_draw = [&](uint64_t it, int ch) {
        double nk = 0.0;
        if (mine) {
            const Draw2 d = draw2(P.seed, it, (uint32_t)ch, (uint32_t)(3 + (lane >> 1)));
            const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(d.q0)));
            double sn, cs;
            mc3_det_sincos2pi(u01(d.q1), &sn, &cs);
            nk = (rad * ((lane & 1) ? sn : cs)) * pstep_k;
        }
        return nk;
    };
    // PH_CHAINWAVE: what was prepared for the next iteration while the other waves computed
    double normal_ahead = 0.0;
    uint64_t normal_ahead_iter = ~0ull;
    double acc_lo = 0.0, acc_hi = 0.0;          // the Metropolis test in the log domain
    double jump_ahead = 0.0;                    // snooker: the next ordinary jump
    uint64_t jump_ahead_iter = ~0ull;
    (void)normal_ahead; (void)normal_ahead_iter; (void)acc_lo; (void)acc_hi;
    (void)jump_ahead; (void)jump_ahead_iter;
    // the iteration that follows (iter, this chain) in this workgroup's order, and its chain;
    // new_refresh: it belongs to the next refresh (whose history holds this refresh's rows)
    auto draws_ahead = [&](uint64_t iter, bool last_of_refresh, int q, bool &new_refresh) {
        uint64_t it2 = iter + 1;
        int c2 = c;
        new_refresh = last_of_refresh;
        if constexpr (MC) {
            if (last_of_refresh) {
                const int l2 = lc + (int)gridDim.x;
                if (q + 1 < mc_turns && l2 < P.nlocal) {        // next chain, same refresh
                    it2 = iter + 1 - (uint64_t)((TH == 1) ? 1 : P.thinning);
                    c2 = P.chain0 + l2;
                    new_refresh = false;
                } else {                                        // first chain, next refresh
                    c2 = P.chain0 + (int)blockIdx.x;
                }
            }
        }
        (void)q;
        if (SNOOKER_AHEAD && !MC && sampler == MC3HIP_SNOOKER)
            return get_draws(batch, P, (long long)it2, c2);
        return make_draws(P, it2, c2, sampler);
    };
    for (long long R = R0; R < R0 + nref && alive; ++R) {
    if constexpr (PERSIST) {
        // (MC: lc is the chain of the last turn here; chain 0 belongs to workgroup 0)
        if (P.test_abort >= 0 && R == P.test_abort && (MC ? blockIdx.x == 0 : lc == 0)) {     // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);             // (all lanes, same word)
            if constexpr (phase_hooked(PHASE))
                give_up = true;                 // (leaves with its workgroup, below)
            else
                return;
        }
    }
    const double *snap = snapshot_of(R);
    const long long zsize = history_of(R);
    const int jfirst = (TH == 1) ? 0 : jb;
    const int jlast = (TH == 1) ? 1 : je;
    for (int q = 0; q < mc_turns; ++q) {        // (MC: the chains of this workgroup in turn)
    if constexpr (MC) {
        lc = (int)blockIdx.x + q * (int)gridDim.x;
        if (lc >= P.nlocal)
            break;
        c = P.chain0 + lc;
        if (wave0) {
            const double *slot = mc_state + q * mc_stride;
            xk = mine ? slot[lane] : 0.0;
            cur = slot[zs];
            best = slot[zs + 1];
            nacc = (long long)mc3_bits(slot[zs + 2]);
        }
    }
    for (int j = jfirst; j < jlast; ++j) {
        if constexpr (WL)
            MC3_LSTAMP(5);
        else
            MC3_STAMP(5);
        unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tkP = 0, tkT = 0, tkC = 0;
        (void)tkP; (void)tkT; (void)tkC;
        MC3_TICK(tk0);
#ifdef MC3HIP_STAMPS
        if (tk_prevend != 0) {
            ta11 += tk_prevend - tk_prev3;   // end of iteration -> row published
            ta13 += tk0 - tk_prevend;        // -> top of the next iteration
        }
#endif
        const uint64_t iter = (TH == 1) ? (uint64_t)R : (uint64_t)(R * P.thinning + j);
        double npk = 0.0, zk = 0.0, u = 0.0;
        double mrfactor = 1.0;
        bool sjump = false, outk = false, fast_anyout = false;
        int flags = 0;
        const Draw2 d0 = nxt.d0, d1 = nxt.d1, d2 = nxt.d2;
        if constexpr (!PERSIST)
            nxt = make_draws(P, iter + 1, c, sampler);
        // Metropolis draw, chain.py:259
        u = (sampler == MC3HIP_SNOOKER) ? u01(d2.q1) : u01(d0.q1);
        if (PHASE != PH_FINISH && wave0) {
            // support-distribution draw, chain.py:185 (the chain wave of PH_CHAINWAVE has
            // drawn it while the likelihood of the iteration before was computed)
            double normal_k = 0.0;
            if (need_normal) {
                if (PHASE == PH_CHAINWAVE && normal_ahead_iter == iter)
                    normal_k = normal_ahead;
                else
                    normal_k = normal_draw(iter, c);
            }
            double jump_k = 0.0;
            if (PHASE == PH_CHAINWAVE && sampler == MC3HIP_SNOOKER && jump_ahead_iter == iter) {
                jump_k = jump_ahead;                               // (formed during the likelihood)
            } else
            if (sampler == MC3HIP_SNOOKER) {                       // chain.py:195-217
                long long iR1 = (long long)bounded(d0.q0, (uint64_t)zsize);
                long long iR2 = 1 + (long long)bounded(d0.q1, (uint64_t)(zsize - 1));
                if (iR2 == iR1)
                    iR2 = 0;
                sjump = u01(d1.q0) < 0.1;
                const long long iz = (long long)bounded(d1.q1, (uint64_t)zsize);
                double z1 = 0.0, z2 = 0.0;
                if constexpr (PERSIST) {
                    // requested during the previous iteration (request_snooker)
                    if constexpr (!SNOOKER_AHEAD) {
                        Draws now;
                        now.d0 = d0;
                        now.d1 = d1;
                        now.d2 = d2;
                        request_snooker(now, zsize);
                    }
                    const double *const addr[3] = {sa[0], sa[1], sa[2]};
                    double val[3] = {sv[0], sv[1], sv[2]};
                    bool ahead_new_refresh = false;
                    if constexpr (PHASE != PH_CHAINWAVE)
                        nxt = draws_ahead(iter, TH == 1 || j + 1 == jlast, q, ahead_new_refresh);
                    alive = wait_rows<3, P2P>(addr, val, ctl_rw);
                    // (the chain wave: these rows arrived during the likelihood before last; a
                    // wait here would only wait for the Z row just published)
                    if constexpr (PHASE != PH_CHAINWAVE)
                        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see lean_demc_kernel
                    z1 = val[0];
                    z2 = val[1];
                    zk = val[2];
                    // the next iteration's rows: its history also holds this refresh's rows
                    // once this iteration is the last of the refresh
                    if (SNOOKER_AHEAD && PHASE != PH_CHAINWAVE && alive)
                        request_snooker(nxt, ahead_new_refresh ? history_of(R + 1) : zsize);
                } else {
                    z1 = mine ? P.Z[(size_t)iR1 * zs + lane] : 0.0;
                    z2 = mine ? P.Z[(size_t)iR2 * zs + lane] : 0.0;
                    zk = (mine && sjump) ? P.Z[(size_t)iz * zs + lane] : 0.0;
                }
                if (sjump) {
                    flags |= 4;
                    const double gs = 1.2 + (2.2 - 1.2) * u01(d2.q0);
                    const bool same = __all(!mine || zk == xk);
                    if (same) {
                        jump_k = gs * (z2 - z1);
                    } else {
                        const double dz = xk - zk;
                        const double zp1 = lane_serial_sum(z1 * dz, nf);
                        const double zp2 = lane_serial_sum(z2 * dz, nf);
                        const double dd = lane_serial_sum(dz * dz, nf);
                        jump_k = ((gs * (zp1 - zp2)) * dz) / dd;
                    }
                } else {
                    jump_k = P.gamma * (z1 - z2) + P.fepsilon * normal_k;
                }
            } else if (sampler == MC3HIP_MRW) {                    // chain.py:219-220
                jump_k = normal_k;
                if constexpr (PERSIST && PHASE != PH_CHAINWAVE) {
                    bool unused;
                    nxt = draws_ahead(iter, TH == 1 || j + 1 == jlast, q, unused);
                }
            } else {                                                 // chain.py:221-232
                int r1, r2;
                demc_partners(d0.q0, P.nchains, c, r1, r2);
                double x1 = 0.0, x2 = 0.0;
                if constexpr (PERSIST) {
                    const int kk = mine ? lane : 0;
                    const double *const addr[2] = {snap + (size_t)r1 * zs + kk,
                                                   snap + (size_t)r2 * zs + kk};
                    double val[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        val[q] = load_published<P2P>(addr[q]);
#ifdef MC3HIP_STAMPS
                    unsigned long long tkL;
                    MC3_TICK(tkL);
                    ta15 += tkL - tk0;
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        bool unused;
                        nxt = draws_ahead(iter, TH == 1 || j + 1 == jlast, q, unused);
                    }
                    pin_scalar(nxt.d0.q0);
                    pin_scalar(nxt.d0.q1);
                    __builtin_amdgcn_sched_barrier(0);
#ifdef MC3HIP_STAMPS
                    alive = wait_rows<2, P2P>(addr, val, ctl_rw, &ta14);
#else
                    alive = wait_rows<2, P2P>(addr, val, ctl_rw);
#endif
                    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see lean_demc_kernel
                    MC3_TICK(tk1);
                    x1 = val[0];
                    x2 = val[1];
                } else {
                    x1 = mine ? snap[(size_t)r1 * zs + lane] : 0.0;
                    x2 = mine ? snap[(size_t)r2 * zs + lane] : 0.0;
                }
                jump_k = P.gamma * (x1 - x2) + P.fepsilon * normal_k;
            }
#ifdef MC3HIP_STAMPS
            if (tk1 == 0)
                MC3_TICK(tk1);               // (samplers without a poll: the jump is known)
#endif
            // proposal and boundaries, chain.py:235-243
            npk = xk + jump_k;
            outk = mine && (npk < pmin_k || npk > pmax_k);
            const bool anyout = __any(outk);
            if constexpr (FAST) {
                fast_anyout = anyout;
            } else {
            if (lane < P.nparams)
                L.nextp[lane] = fixed0;
            if (lane + 64 < P.nparams)
                L.nextp[lane + 64] = fixed1;
            __builtin_amdgcn_wave_barrier();
            if (mine)
                L.nextp[jfree] = npk;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                for (int s = 0; s < P.nshare; ++s)                   // chain.py:246-247
                    L.nextp[P.ishare[s]] = L.nextp[P.share_src[s]];
                L.flag[0] = anyout ? 1 : 0;
                L.flag[2] = alive ? 1 : 0;
            }
            if constexpr (phase_hooked(PHASE)) {
                __builtin_amdgcn_wave_barrier();
                hook->publish(P, lc, L.nextp);
                MC3_TICK(tkP);
            }
            }
            if constexpr (PHASE == PH_PROPOSE) {
                // hand the proposal to the tile and finish kernels
                if (sjump && !anyout) {                              // chain.py:251-255
                    const double dc = xk - zk, dn = npk - zk;
                    const double cnorm = lane_serial_sum(mine ? dc * dc : 0.0, nf);
                    const double nnorm = lane_serial_sum(mine ? dn * dn : 0.0, nf);
                    mrfactor = mc3_det_pow_half(nnorm / cnorm, nf - 1);
                }
                __builtin_amdgcn_wave_barrier();
                for (int q = lane; q < P.nparams; q += 64)
                    P.sc_nextp[(size_t)lc * SC_STRIDE + q] = L.nextp[q];
                if (mine)
                    P.sc_npk[(size_t)lc * nf + lane] = npk;
                if (lane == 0) {
                    P.sc_scal[2 * lc] = u;
                    P.sc_scal[2 * lc + 1] = mrfactor;
                    P.sc_flag[lc] = (anyout ? 1 : 0) | (sjump ? 2 : 0);
                }
                if (outk)
                    atomicAdd(&P.outbounds[lane], 1ull);             // chain.py:242-243
                return;
            }
        }
        if constexpr (PHASE == PH_FINISH) {
            // resume: the proposal and its scalars come back from global scratch
            const int fl = P.sc_flag[lc];
            npk = mine ? P.sc_npk[(size_t)lc * nf + lane] : 0.0;
            u = P.sc_scal[2 * lc];
            mrfactor = P.sc_scal[2 * lc + 1];
            sjump = (fl & 2) != 0;
            if (sjump)
                flags |= 4;
            if (tid == 0) {
                L.flag[0] = fl & 1;
                L.flag[2] = 1;
            }
        }
        bool anyout;
        if constexpr (FAST) {
            anyout = fast_anyout;
            if (PERSIST && !alive)
                break;
        } else {
            block_sync<WPC>();
            if constexpr (WL)
                MC3_LSTAMP(2);
            else
                MC3_STAMP(2);
            anyout = L.flag[0] != 0;
            if constexpr (phase_hooked(PHASE)) {
                // the tile phase of this iteration, with the whole workgroup; false for all of
                // its waves once one of them has given up
                if constexpr (PHASE == PH_CHAINWAVE) {
                    hook->skip = anyout;       // (nothing to compute for a proposal out of bounds)
                    // what the chain wave does while the other waves compute the likelihood:
                    // everything of the NEXT iteration that needs no chain state, and the
                    // logarithm this iteration's Metropolis test will be decided with
                    const bool last_j = (TH == 1 || j + 1 == jlast);
                    auto prepare = [&]() {
                        accept_window(u, acc_lo, acc_hi);
                        if (sampler != MC3HIP_DEMC) {
                            bool new_refresh;
                            nxt = draws_ahead(iter, last_j, q, new_refresh);
                            if (sampler == MC3HIP_SNOOKER)
                                request_snooker(nxt, new_refresh ? history_of(R + 1) : zsize);
                        }
                        if (need_normal) {
                            normal_ahead = normal_draw(iter + 1, c);
                            normal_ahead_iter = iter + 1;
                        }
                        if (sampler == MC3HIP_SNOOKER) {
                            // the next iteration's jump itself, when it is an ordinary one
                            // (chain.py:216-217; nine in ten) and its two rows are there: it
                            // needs no chain state, and the likelihood takes longer than the
                            // rows' round trip
                            jump_ahead_iter = ~0ull;
                            if (!(u01(nxt.d1.q0) < 0.1)) {
                                __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
                                const bool there = !__any(mc3_bits(sv[0]) == ROW_PENDING ||
                                                          mc3_bits(sv[1]) == ROW_PENDING);
                                if (there) {
                                    jump_ahead = P.gamma * (sv[0] - sv[1]) +
                                                 P.fepsilon * (need_normal ? normal_ahead : 0.0);
                                    jump_ahead_iter = iter + 1;
                                }
                            }
                        }
                    };
                    alive = hook->tile_phase(P, (L.flag[2] != 0) && !give_up, prepare);
                } else
                alive = hook->tile_phase(P, (L.flag[2] != 0) && !give_up);
                MC3_TICK(tkT);
                if (!alive)
                    break;
            } else if constexpr (PERSIST) {
                alive = L.flag[2] != 0;
                if (!alive)
                    break;                     // a wait timed out: stop this chain
            }
        }
        double tile_total = 0.0;
        if constexpr (phase_hooked(PHASE))
            tile_total = hook->collect(P, lc);  // (also of a proposal that is out of bounds:
                                                // the sums must be seen before the row is reused)
        (void)tile_total;
        MC3_TICK(tkC);
        double nextchisq = mc3_from_bits(0x7ff8000000000000ull);
        if (!anyout) {
            flags |= 1;
            if constexpr (FAST) {
                Model<MT> model;
                if constexpr (!FX) {
                    model.init_from_lanes(P, npk);             // every coefficient is free
                } else {
                    // FX: some coefficients are fixed and/or carry Gaussian priors (a separate
                    // instantiation, so that the plain kernels stay exactly as they were tuned)
                    model.init_from_lanes(P, npk, kf, fx);
                }
                nextchisq = block_gauss_model<WPC, MT, MULTI, MULTI>(P, model, L.part, L.flag, tile,
                                                                     &tile1);
                if constexpr (FX)
                    nextchisq = nextchisq + priors.term(npk, lane);
            } else if constexpr (PHASE == PH_FINISH) {
                nextchisq = finish_chisq<MT>(P, lc);
            } else if constexpr (phase_hooked(PHASE)) {
                nextchisq = hook->template finish<MT>(P, tile_total, L.nextp);
            } else {
                nextchisq = block_chisq<WPC, MT, WL, MULTI>(P, L.nextp, L.part, L.flag, L.tab,
                                                            L.halo, L.xs, tile);   // chain.py:249
            }
            if constexpr (WL)
                MC3_LSTAMP(3);
            else
                MC3_STAMP(3);
            MC3_TICK(tk2);
            if (wave0) {
                if (PHASE != PH_FINISH && sjump) {                   // chain.py:251-255
                    const double dc = xk - zk, dn = npk - zk;
                    const double cnorm = lane_serial_sum(mine ? dc * dc : 0.0, nf);
                    const double nnorm = lane_serial_sum(mine ? dn * dn : 0.0, nf);
                    mrfactor = mc3_det_pow_half(nnorm / cnorm, nf - 1);
                }
                bool accepted;
                if (PHASE == PH_CHAINWAVE && !sjump)                 // (mrfactor = 1 exactly)
                    accepted = metropolis_accept(0.5 * (cur - nextchisq), u, acc_lo, acc_hi);
                else
                    accepted = mc3_det_exp(0.5 * (cur - nextchisq)) * mrfactor > u;
                if (accepted) {                                      // chain.py:257-274
                    flags |= 2;
                    xk = npk;
                    cur = nextchisq;
                    nacc += 1;
                    if (nextchisq < best) {
                        best = nextchisq;
                        if (mine)
                            P.bestx[(size_t)lc * nf + lane] = xk;
                        if (lane == 0)
                            P.best_iter[lc] = (long long)iter;
                    }
                }
            }
        } else if (PHASE != PH_FINISH && outk) {
            atomicAdd(&P.outbounds[lane], 1ull);                     // chain.py:242-243
        }
        if (!FAST && P.trace_iters > 0 && wave0 && lane == 0) {
            const long long trow = (R - P.ctl[1]) * P.thinning + j;
            if (trow >= 0 && trow < P.trace_iters) {
                P.trace_chisq[trow * P.nlocal + lc] = nextchisq;
                P.trace_flag[trow * P.nlocal + lc] = flags;
            }
        }
        if constexpr (SNOOKER_AHEAD) {
            // the rows requested ahead have long arrived: saying so before the publish keeps
            // the wait for them from landing behind the stores (see lean_demc_kernel)
            if (sampler == MC3HIP_SNOOKER)
                __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        if constexpr (!FAST)
            block_sync<WPC>();
        if constexpr (WL) {
            MC3_LSTAMP(6);
            MC3_LSTAMP_FLUSH();
        } else {
            MC3_STAMP(6);
        }
#ifdef MC3HIP_STAMPS
        MC3_TICK(tk3);
        if (tk1 != 0 && tk2 != 0) {
            ta8 += tk1 - tk0;      // top -> partner rows arrived
            ta9 += tk2 - tk1;      // -> chisq reduced
            ta10 += tk3 - tk2;     // -> end of iteration (accept; publish is after)
            ta12 += 1;
            if constexpr (phase_hooked(PHASE)) {
                tr0 += tkP - tk1;  // rows arrived -> proposal published
                tr1 += tkT - tkP;  // -> tile phase done (waits for the column's other owners)
                tr2 += tkC - tkT;  // -> all tiles' sums collected
                tr3 += tk2 - tkC;  // -> chisq known
            }
        }
        tk_prev3 = tk3;
#endif
    }
    if (!alive)
        break;
    // Z append, chain.py:276-289 (strided slot M0 + ID + k*nchains)
    if (wave0 && (TH == 1 || je == P.thinning)) {
        const long long row = P.M0 + R * P.nchains + c;
        if constexpr (P2P)
            store_row_everywhere(P, peers_z, (size_t)row * zs + lane, xk, mine);   // publish
        if (mine) {
            if constexpr (P2P) {
            } else if constexpr (PERSIST) {
                __hip_atomic_store(&P.Z[(size_t)row * zs + lane], xk, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);      // publish (sc1)
            } else {
                P.Z[(size_t)row * zs + lane] = xk;
            }
        }
        if (lane == 0) {
            if constexpr (P2P) {
                peers_lp.store((size_t)row, -0.5 * cur);
            } else {
                P.log_post[row] = -0.5 * cur;
            }
            P.zchain[row] = c;
        }
    }
    if constexpr (MC) {
        if (wave0) {                              // this chain waits for its next turn
            double *slot = mc_state + q * mc_stride;
            if (mine)
                slot[lane] = xk;
            if (lane == 0) {
                slot[zs] = cur;
                slot[zs + 1] = best;
                slot[zs + 2] = mc3_from_bits((unsigned long long)nacc);
            }
        }
    }
    }   // turns
    MC3_TICK(tk_prevend);
    }   // refreshes
    if constexpr (MC) {
        if (wave0) {
            for (int q = 0; q < mc_turns; ++q) {
                const int l = (int)blockIdx.x + q * (int)gridDim.x;
                if (l >= P.nlocal)
                    break;
                const double *slot = mc_state + q * mc_stride;
                if (mine)
                    P.X[(size_t)l * nf + lane] = slot[lane];
                if (lane == 0) {
                    P.chisq[l] = slot[zs];
                    P.best_chisq[l] = slot[zs + 1];
                    P.numaccept[l] = (long long)mc3_bits(slot[zs + 2]);
                }
            }
        }
    } else if (wave0) {
        if (mine)
            P.X[(size_t)lc * nf + lane] = xk;
        if (lane == 0) {
            P.chisq[lc] = cur;
            P.best_chisq[lc] = best;
            P.numaccept[lc] = nacc0 + nacc;
        }
    }
    MC3_STAMP(4);
#ifdef MC3HIP_STAMPS
    if (PERSIST && (phase_hooked(PHASE) ? (lane == 0 && lc == 0 && blockIdx.y == 0 &&
                                           (PHASE == PH_CHAINWAVE || threadIdx.x == 0))
                                        : (threadIdx.x == 0 && blockIdx.x == 0))) {
        // (the chain wave of PH_CHAINWAVE: second page -- the compute waves' LDS stamps own 0..15)
        constexpr int TB = (PHASE == PH_CHAINWAVE) ? 16 : 0;
        P.stamps[TB + 8] += ta8; P.stamps[TB + 9] += ta9; P.stamps[TB + 10] += ta10;
        P.stamps[TB + 11] += ta11; P.stamps[TB + 12] += ta12; P.stamps[TB + 13] += ta13;
        P.stamps[TB + 14] += ta14; P.stamps[TB + 15] += ta15;
        if constexpr (phase_hooked(PHASE)) {
            P.stamps[16] += tr0; P.stamps[17] += tr1; P.stamps[18] += tr2; P.stamps[19] += tr3;
        }
    }
#endif
}

// one Z refresh per launch (also the multi-GPU path: a collective follows each launch)
// (multi-GPU, graph-captured exchange) this chain's new row also goes into slot `roff` of
// the exchange ring, where the captured all-gather finds it at a fixed address.  The row of
// a refresh is the chain's state at its end, which run_chain has just stored to P.X from
// these very lanes.
__device__ __forceinline__ void ring_out(const DevProblem &P, int roff)
{
    if (P.xring != nullptr && threadIdx.x < (unsigned)P.nfree)
        P.xring[((size_t)roff * P.nchains + P.chain0 + blockIdx.x) * P.zs + threadIdx.x] =
            P.X[(size_t)blockIdx.x * P.nfree + threadIdx.x];
}

}  // namespace mc3hip
