// kernels_lean.h -- the lean one-wave persistent kernels.
#pragma once
#include "run_chain.h"

namespace mc3hip {

// the lean instance (see run_chain): one wave per chain, MT free polynomial
// coefficients, Gaussian chisq, ndata <= 1024, no priors/shared/fixed/epsilon/trace
// (MULTI: more than 1024 points; the first tile stays in registers, the rest is streamed)
// (MC: more chains than co-resident workgroups -- a wave advances several, run_chain MC)
template <int MT, int SMP, bool P2P = false, int TH = 0, bool FX = false, bool MULTI = false,
          bool MC = false>
__global__ __launch_bounds__(64) void persist_fast_kernel(const DevProblem P, const int nref)
{
    run_chain<1, MT, false, MULTI, true, SMP, true, PH_FUSED, P2P, TH, FX, NoHook, MC>(
        P, P.ctl[0], nref);
}

// ---------------------------------------------------------------------------------
// The headline shape written out by hand: lean instance (see run_chain FAST), DEMC,
// thinning = 1, persistent.  Same arithmetic and the same results as
// run_chain<1, MT, false, false, true, MC3HIP_DEMC, true>; what differs is only how the
// loop is laid out for ONE wave that nothing else on its SIMD hides latencies for:
//   * every address is a pointer bumped by a constant per refresh (one 128-byte line per
//     row), no 64-bit index arithmetic in the loop;
//   * everything of refresh R+1 that does not depend on chain states -- Philox, the
//     Metropolis draw, the partner indices and their byte offsets -- is computed in the
//     shadow of refresh R's partner-row wait;
//   * the Metropolis update is a select, not a branch; only the rare events (new best
//     fit, out-of-bounds proposal, a failed wait) leave the straight line.
//
// BATCH (ensembles of more than one wave per SIMD, where the loop is bound by the instructions
// it issues, not by the hand-off): the draws of 64 consecutive refreshes are computed at once,
// one refresh per lane -- Philox, log u and its window, the partners' row offsets as vector
// instructions once per 64 refreshes instead of ~100 scalar and ~100 vector instructions per
// refresh -- and a refresh reads its own with v_readlane.  Same draws, same results.
//
// COURIER (peer-to-peer mode, several GPUs): the rows for the OTHER ranks' replicas are stored
// by a second wave of the workgroup.  A wave's loads and stores retire through one counter in
// issue order (MI355X_MICROARCH.md, s_waitcnt): a poll of the next refresh's partner rows,
// issued behind the stores into seven remote replicas, would have its data held back until
// every one of those stores has been acknowledged over xGMI -- a full round trip, where the
// hand-off itself needs one way.  So the chain wave stores into the LOCAL replica only and
// leaves the row in an LDS ring; the courier picks it up (it spins on an LDS counter: no
// vector-memory operation, nothing the chain wave could wait behind) and stores it into the
// remote replicas, its own counter absorbing the acknowledgements.  Same rows, same results.
//
// EPS (fepsilon != 0, chain.py:185,232: the support distribution's draw added to the jump): lane
// k's normal deviate of the NEXT refresh -- Philox slot 3 + k/2, a logarithm, a square root, a
// sine or cosine, as run_chain's normal_draw -- is computed on the vector unit in the same
// shadow of the partner rows' round trip as the scalar draws.
// (NW = 2, 4: helper waves share the data sum -- gauss.h SumHelpers; run-time specialised
// design matrices whose tile one wave's registers do not hold)
// (NT = 2, 4, 8 with NW = 4: more than 1024 points in the registers of four waves -- what
// lean_demc_wide_kernel does for the plain shape, here for fixed / shared coefficients, priors,
// the support distribution; specialised at run time, round 5)
// BIG (the plain shape with the draws of 64 refreshes at once, more than 1 024 chains: two and more
// waves per SIMD): the wait for the partners' rows laid out by hand and a longer pause in front
// of the polls -- see the wait below.
template <int MT, bool P2P, bool FX, bool MULTI, bool BATCH, bool COURIER, bool EPS, int NW = 1,
          int NT = 1, bool BIG = false>
__device__ __forceinline__ void lean_demc_body(const DevProblem &P, const int nref)
{
    static_assert(!BIG || (BATCH && !P2P && !FX && !MULTI && !COURIER && NW == 1), "BIG: the plain shape");
    static_assert(NT == 1 || NW > 1, "several tiles: in the registers of several waves");
    static_assert(!EPS || (!P2P && !MULTI && !BATCH && !COURIER), "EPS: the one-GPU headline loop");
    static_assert(NW == 1 || (!P2P && !MULTI && !COURIER), "helper waves: one GPU, one tile");
    constexpr int NC = Model<MT>::NC;         // coefficients the model tag fixes
    const int nf = FX ? P.nfree : NC;        // FX: some coefficients are fixed (nfree < MT)
    constexpr int ZS = ZROW;                 // doubles per Z row: one 128-byte line (NF <= 16)
    constexpr int LM = P2P ? 1 : 0;
    const int lane = (COURIER || NW > 1) ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const int lc = blockIdx.x;
    const int c = P.chain0 + lc;
    const bool mine = lane < nf;
    const int kk = mine ? lane : 0;
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    // (the ring: the courier's, or -- test hook P.remote_delay without a courier -- the chain
    // wave's own queue of rows whose time has not come, handoff.h DelayedRows)
    __shared__ double s_ring[(COURIER || P2P) ? COURIER_RING * ZROW : 1];
    __shared__ int s_cour[4];                // 1: rows sent (every eighth), 2: done, 3: go
    (void)s_ring; (void)s_cour;
    __shared__ double s_hpart[NW > 1 ? 64 * NW : 1], s_hcf[16];
    __shared__ int s_hcmd[2];
    (void)s_hpart; (void)s_hcf; (void)s_hcmd;
    SumHelpers<MT, (NW > 1 ? NW : 2), NT> helpers;
    (void)helpers;
    if constexpr (COURIER) {
        if (!courier_begin(s_ring, s_cour, ctl_rw))
            return;
        if (threadIdx.x >= 64) {
            courier_run<P2P>(P, s_ring, s_cour, lane, nf, c);
            return;
        }
    } else if constexpr (NW > 1) {
        if (!helpers.enter(s_hpart, s_hcf, s_hcmd,
                           __hip_atomic_load(&ctl_rw[CTL_ABORT], __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT) != 0))
            return;
    } else {
        if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(
                &ctl_rw[CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
            return;
    }
    // (NT tiles of 1024 points in the registers of the NW waves between them; tiles beyond the
    // data: weight 0)
    DataTile<NW, MT> tiles[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        if (q == 0 || q * VLANES < P.npad) {
            tiles[q].load(P, q * VLANES);
        } else if constexpr (NT > 1) {
            tiles[q].clear();
        }
    }
    const DataTile<NW, MT> &tile = tiles[0];
    if constexpr (NW > 1) {
        if (threadIdx.x >= 64) {
            helpers.serve(P, tiles);
            return;
        }
    }
    int handed = 0;                          // COURIER: rows left in the ring so far
    (void)handed;
    PeerTable<double> peers_z;
    PeerTableLazy<double> peers_lp;      // (the log_post columns: rare, handoff.h)
    if constexpr (P2P) {
        if constexpr (!COURIER)
            peers_z.load(P.peerZ, P.npeers);
        peers_lp.load(P.peerLP, P.npeers);
    }
    (void)peers_z; (void)peers_lp;
    DelayedRows late;
    late.begin(P, P.ctl[0], c, lane);
    (void)late;
    MC3_STAMP(0);
    DataTile<1, MT> tile1;                   // MULTI: points 1024..2047 stay in registers too
    if constexpr (MULTI)
        tile1.load(P, VLANES);
    const long long R0 = P.ctl[0];
    const int n = P.nchains;
    const size_t blockstride = (size_t)n * ZS;                   // doubles per refresh
    // rows of refresh R0 - 1 (the current states) and of refresh R0 (to be published)
    const double *pbase = (R0 == 0) ? P.snap0 : P.Z + (size_t)(P.M0 + (R0 - 1) * n) * ZS;
    double *zblock = P.Z + (size_t)(P.M0 + R0 * n) * ZS;
    double *lp_row = P.log_post + (P.M0 + R0 * n + c);
    int32_t *zc_row = P.zchain + (P.M0 + R0 * n + c);
    double xk = mine ? pbase[(size_t)c * ZS + lane] : 0.0;
    double cur = P.chisq[lc];
    double best = P.best_chisq[lc];
    const long long nacc0 = P.numaccept[lc];
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    const double gamma = P.gamma, feps = P.fepsilon;
    const long long abort_at = (P.test_abort >= 0 && lc == 0) ? (long long)P.test_abort : -1;
    int nacc = 0;
    bool alive = true;
    int kf[NC];
    double fx[NC];
    PriorLanes<NC> priors;
    int hkf = 0;                               // (helper waves, FX: what coefficient `lane` is)
    double hfx = 0.0;
    if constexpr (FX) {
        coefficient_map<NC>(P, kf, fx);
        priors.init(P, kf, fx, lane);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            hkf = (lane == j) ? kf[j] : hkf;
            hfx = (lane == j) ? fx[j] : hfx;
        }
    }
    (void)kf; (void)fx; (void)priors; (void)hkf; (void)hfx;
    // log_post / zchain of the last (up to) 64 refreshes wait in lane (R - R0) mod 64 and
    // are written 64 at a time: nobody reads them inside the launch
    double lpbuf = 0.0;
    MC3_STAMP(1);

    // EPS: the support distribution's draws (chain.py:185) for 32 refreshes at once -- lane
    // 2 i + s: Philox slot 3 + s of refresh nbase + i, i.e. the deviates of parameters 2 s and
    // 2 s + 1 (rad cos, rad sin) -- ~250 vector instructions per batch instead of per refresh
    // (drawn every refresh they did not fit the shadow of the partner rows' round trip: 156 M
    // chain-steps/s against 217 without epsilon); a refresh gathers its own through the crossbar
    // (more than four free parameters -- models specialised at run time: ES = 4 / 8 slots of
    // deviates per refresh, 16 / 8 refreshes at once; lane ES i + s: slot 3 + s of refresh
    // nbase + i)
    constexpr int ES = Model<MT>::NC <= 4 ? 2 : (Model<MT>::NC <= 8 ? 4 : 8);
    constexpr int ESH = ES == 2 ? 1 : (ES == 4 ? 2 : 3);
    const double pstep_k = (EPS && mine) ? P.fstep[lane] : 0.0;
    double ncs = 0.0, nsn = 0.0;
    long long nbase = R0;
    auto fill_normals = [&](long long base) {
        const Draw2 dn = draw2(P.seed, (uint64_t)(base + (lane >> ESH)), (uint32_t)c,
                               (uint32_t)(3 + (lane & (ES - 1))));
        const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(dn.q0)));
        double sn, cs;
        mc3_det_sincos2pi(u01(dn.q1), &sn, &cs);
        ncs = rad * cs;
        nsn = rad * sn;
        nbase = base;
    };
    auto normal_draw = [&](uint64_t it) {
        if (__builtin_expect((long long)it - nbase >= 64 / ES, 0))
            fill_normals((long long)it);
        const int idx = ES * (int)((long long)it - nbase);
        const int src = mine ? idx + (lane >> 1) : idx;
        const double gc = lane_gather(ncs, src), gn = lane_gather(nsn, src);
        return mine ? ((lane & 1) ? gn : gc) * pstep_k : 0.0;
    };
    (void)ncs; (void)nsn; (void)nbase;
    // draws of the first refresh
    Draw2 d = draw2(P.seed, (uint64_t)R0, (uint32_t)c, 0u);
    double u = u01(d.q1);
    // the Metropolis test in the log domain, decided ahead of time: see accept_window
    double acc_lo, acc_hi;
    accept_window(u, acc_lo, acc_hi);
    int r1, r2;
    demc_partners(d.q0, n, c, r1, r2);
    int off1 = r1 * ZS + kk, off2 = r2 * ZS + kk;
    // (behind the wave-uniform draws of the same iteration: the two Philox calls share their
    // first round, which must be computed where it is uniform -- the compiler would otherwise
    // place it under the lanes' mask and could not hand it to the scalar unit any more)
    pin_scalar(d.q0);
    pin_scalar(d.q1);
    double normal_k = 0.0;
    if constexpr (EPS) {
        fill_normals(R0);
        normal_k = normal_draw((uint64_t)R0);
    }
    (void)normal_k; (void)pstep_k;
    // BATCH: lane i holds the draws of refresh bbase + i
    double bu = 0.0, blo = 0.0, bhi = 0.0;
    int bo1 = 0, bo2 = 0;
    long long bbase = R0;
    auto fill = [&](long long base) {
        const Draw2 dd = draw2(P.seed, (uint64_t)(base + lane), (uint32_t)c, 0u);
        bu = u01(dd.q1);
        accept_window(bu, blo, bhi);
        int q1, q2;
        demc_partners(dd.q0, n, c, q1, q2);
        bo1 = q1 * ZS;
        bo2 = q2 * ZS;
        bbase = base;
    };
    if constexpr (BATCH)
        fill(R0);
    (void)bu; (void)blo; (void)bhi; (void)bo1; (void)bo2; (void)bbase;

    unsigned long long tk_prev3 = 0, tk_prevend = 0;
    unsigned long long ta8 = 0, ta9 = 0, ta10 = 0, ta11 = 0, ta12 = 0, ta13 = 0, ta14 = 0, ta15 = 0;
    (void)tk_prev3; (void)tk_prevend; (void)ta8; (void)ta9; (void)ta10; (void)ta11; (void)ta12;
    (void)ta13; (void)ta14; (void)ta15;
    const long long Rend = R0 + nref;
    long long rdone = R0;                    // refreshes completed (published) by this launch
    auto flush_columns = [&](int count) {
        if (lane < count) {
            if constexpr (P2P) {
                const size_t roff = (size_t)(lp_row - P.log_post) + (size_t)lane * n;
                peers_lp.store(roff, lpbuf);
            } else {
                lp_row[(size_t)lane * n] = lpbuf;
            }
            zc_row[(size_t)lane * n] = c;
        }
    };
    // one GPU, no courier: the row is published where it is decided (below), not at the end of
    // the refresh's bookkeeping
    constexpr bool EARLY = !P2P && !COURIER;
    auto publish_early = [&](const double row_k) {
        // (every poll has long returned -- said here for the compiler's wait-count bookkeeping,
        // as at the end of the refresh)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (mine)
            __hip_atomic_store(zblock + (size_t)c * ZS + lane, row_k, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);            // publish (sc1)
    };
    (void)publish_early;
    // every load issued so far is complete before the loop is entered, so that no path
    // through it carries a pending load to the latch (see the note after wait_rows)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (long long R = R0; R < Rend; ++R) {
        unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
        (void)tk0; (void)tk1; (void)tk2; (void)tk3;
        MC3_TICK(tk0);
#ifdef MC3HIP_STAMPS
        if (tk_prevend != 0) {
            ta11 += tk_prevend - tk_prev3;
            ta13 += tk0 - tk_prevend;
        }
#endif
        if (__builtin_expect(R == abort_at, 0)) {                                         // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);             // (all lanes, same word)
            break;            // (the host restores its checkpoint after an abort)
        }
        // partner rows of this refresh (chain.py:221-232), requested first
        // (a short pause lets the stores of the previous refresh clear the memory pipeline
        // ahead of the polls: +3 % measured, 64..192 cycles all the same)
        // (BIG: 320 cycles -- 2 048 chains 1 364 / 1 383 / 1 390 / 1 399 / 1 410 / 1 399 M
        // chain-steps/s for 0 / 64 / 128 / 192 / 320 / 512 cycles with the wait below; 256
        // chains lose with every cycle beyond 64)
        __builtin_amdgcn_s_sleep(BIG ? 5 : 1);
        if constexpr (BATCH) {
            const int idx = (int)(R - bbase);
            off1 = __builtin_amdgcn_readlane(bo1, idx) + kk;
            off2 = __builtin_amdgcn_readlane(bo2, idx) + kk;
            u = read_lane(bu, idx);
            acc_lo = read_lane(blo, idx);
            acc_hi = read_lane(bhi, idx);
        }
        const double *const addr[2] = {pbase + off1, pbase + off2};
        double val[2];
        constexpr int KR = 1;            // polls kept in flight: more measured slower (poll_ring)
        double ring[KR][2];
        ring[0][0] = load_published<LM>(addr[0]);
        ring[0][1] = load_published<LM>(addr[1]);
#ifdef MC3HIP_STAMPS
        unsigned long long tkL;
        MC3_TICK(tkL);
        ta15 += tkL - tk0;
#endif
        const double u_now = u, lo_now = acc_lo, hi_now = acc_hi;
        const double normal_now = normal_k;
        (void)normal_now;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BATCH) {
            // ... and behind them, once in 64 refreshes, the draws of the next 64
            if (__builtin_expect(R - bbase == 63, 0)) {
                fill(R + 1);
                asm volatile("; pin" : "+v"(bu), "+v"(blo), "+v"(bhi), "+v"(bo1), "+v"(bo2));
            }
        } else {
        // ... and behind them everything of the next refresh that needs no chain state
        d = draw2(P.seed, (uint64_t)(R + 1), (uint32_t)c, 0u);
        pin_scalar(d.q0);
        pin_scalar(d.q1);
        u = u01(d.q1);                                               // chain.py:259
        accept_window(u, acc_lo, acc_hi);
        demc_partners(d.q0, n, c, r1, r2);
        off1 = r1 * ZS + kk;
        off2 = r2 * ZS + kk;
        asm volatile("; pin" : "+v"(u), "+v"(off1), "+v"(off2), "+v"(acc_lo), "+v"(acc_hi));
        if constexpr (EPS) {
            normal_k = normal_draw((uint64_t)(R + 1));
            asm volatile("; pin" : "+v"(normal_k));
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        // (test hook, no courier: the rows held back keep going out while this wave waits)
        auto idle = [&]() {
            if constexpr (P2P && !COURIER) {
                if (__builtin_expect(P.remote_delay != 0, 0))
                    late.drain(P, s_ring, peers_z, lane, mine, false);
            }
        };
        if constexpr (BIG) {
            // The wait, laid out by hand.  In the compiler's layout of poll_ring's two nested
            // loops three taken branches and six register copies lie between the arrival of the
            // later row and the proposal (a taken branch costs a wave ~13 ns,
            // profiles/r06_fp64_operand_probe.txt).  Here the rows are used from the registers
            // they arrive in, both rows there at the first look is the straight line (no branch
            // taken), and the loop that asks again -- only for what is still missing, at once --
            // has one way out, its end.  With two and more waves per SIMD this is worth 2-4 %
            // (2 048 chains: 1 354 -> 1 383, 1 410 with the longer pause above); with one wave per
            // SIMD it LOSES 1 % (256 chains: 227.1 -> 224.5 whatever the pause): there the
            // ensemble's period answers to when the polls are sent, not to what follows them,
            // and those chains keep poll_ring.
            static_assert(KR == 1, "one poll in flight");
            bool pend0 = __any(mc3_bits(ring[0][0]) == ROW_PENDING);
            bool pend1 = __any(mc3_bits(ring[0][1]) == ROW_PENDING);
            if (__builtin_expect(pend0 || pend1, 0)) {
                unsigned spins = 0;
                do {
                    if (pend0)
                        ring[0][0] = load_published<LM>(addr[0]);
                    if (pend1)
                        ring[0][1] = load_published<LM>(addr[1]);
#ifdef MC3HIP_STAMPS
                    ta14 += 1;
#endif
                    if (__builtin_expect((++spins & 255u) == 0u, 0)) {
                        // (the launch given up elsewhere, or nothing for seconds: poll_ring's rule)
                        const int ab = __builtin_amdgcn_readfirstlane(
                            (int)(__hip_atomic_load(&ctl_rw[CTL_ABORT], __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT) != 0));
                        if (ab != 0 || spins > (1u << 22)) {
                            if (ab == 0)
                                __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                            alive = false;                 // (one way out of the loop: its end)
                        }
                    }
                    pend0 = __any((int)alive & (int)(mc3_bits(ring[0][0]) == ROW_PENDING));
                    pend1 = __any((int)alive & (int)(mc3_bits(ring[0][1]) == ROW_PENDING));
                } while (pend0 || pend1);
            }
            val[0] = ring[0][0];
            val[1] = ring[0][1];
        } else {
#ifdef MC3HIP_STAMPS
            alive = poll_ring<KR, LM>(addr, ring, val, ctl_rw, &ta14, idle);
#else
            alive = poll_ring<KR, LM>(addr, ring, val, ctl_rw, nullptr, idle);
#endif
        }
        MC3_TICK(tk1);
        if (__builtin_expect(!alive, 0))
            break;
        // proposal and boundaries, chain.py:232-243
        const double jump_k = gamma * (val[0] - val[1]) + feps * (EPS ? normal_now : 0.0);
        const double npk = xk + jump_k;
        const bool outk = mine && (npk < pmin_k || npk > pmax_k);
        if (__builtin_expect(!__any(outk), 1)) {
            // (helper waves: woken first -- lane j hands them coefficient j, free parameter j or
            // what FX maps there -- so that they work while this wave makes its own copy)
            if constexpr (NW > 1) {
                // (the gather by EVERY lane, then the choice: inside the conditional it would
                // run with the fixed coefficients' lanes switched off, and a lane that is off
                // hands ds_bpermute a zero -- found by tools/fuzz_hunt.py)
                const double got = lane_gather(npk, hkf < 0 ? 0 : hkf);
                helpers.order(FX ? (hkf < 0 ? hfx : got) : npk, lane);
            }
            Model<MT> model;
            if constexpr (!FX) {
                model.init_from_lanes(P, npk);                 // every coefficient is free
            } else {
                model.init_from_lanes(P, npk, kf, fx);         // FX: some are fixed
            }
            double nextchisq;                                          // chain.py:249
            if constexpr (NW == 1) {
                nextchisq = block_gauss_model<1, MT, MULTI, MULTI>(P, model, nullptr, nullptr,
                                                                   tile, &tile1);
            } else {
                double prior = 0.0;
                if constexpr (FX)
                    prior = priors.term(npk, lane);
                nextchisq = helpers.collect(P, model, tiles, lane);
                if constexpr (FX)
                    nextchisq = nextchisq + prior;
            }
            if constexpr (FX && NW == 1)
                nextchisq = nextchisq + priors.term(npk, lane);
            MC3_TICK(tk2);
            const bool acc = metropolis_accept(0.5 * (cur - nextchisq), u_now, lo_now,
                                               hi_now);             // chain.py:257-274
            xk = acc ? npk : xk;
            // (one GPU: the row leaves the moment it is decided -- the best-fit test, the
            // counters and the bookkeeping columns are nobody's critical path, the row is its
            // two partners')
            if constexpr (EARLY)
                publish_early(xk);
            if (__builtin_expect(__any(acc && nextchisq < best), 0)) {
                best = nextchisq;
                if (mine)
                    P.bestx[(size_t)lc * nf + lane] = xk;
                if (lane == 0)
                    P.best_iter[lc] = R;
            }
            cur = acc ? nextchisq : cur;
            nacc += acc ? 1 : 0;
        } else {
            if constexpr (EARLY)
                publish_early(xk);                                   // (the state as it was)
            if (outk)
                atomicAdd(&P.outbounds[lane], 1ull);                 // chain.py:242-243
        }
#ifdef MC3HIP_STAMPS
        MC3_TICK(tk3);
        if (tk2 != 0) {
            ta8 += tk1 - tk0;
            ta9 += tk2 - tk1;
            ta10 += tk3 - tk2;
            ta12 += 1;
        }
        tk_prev3 = tk3;
#endif
        // Every poll has long returned.  Saying so here keeps the compiler's wait-count
        // bookkeeping from guarding the poll registers at the loop latch, where a wait would
        // stall on the acknowledgement of the row about to be published (loads and stores
        // share the counter on gfx9).
        // (EARLY: said before the row left -- here it would wait for the row's acknowledgement)
        if constexpr (!EARLY)
            __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0), already true
        // Z append, chain.py:276-289: publish the row ...
        if constexpr (COURIER) {
            // ... into the local replica, and hand it to the courier for the others
            if constexpr (P2P) {
                if (mine)
                    __hip_atomic_store(zblock + (size_t)c * ZS + lane, xk, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
            }
            alive = courier_hand(s_ring, s_cour, handed, lane, mine, xk, ctl_rw);
            if (__builtin_expect(!alive, 0))
                break;
        } else if constexpr (P2P) {
            const size_t zoff = (size_t)(zblock - P.Z) + (size_t)c * ZS + lane;
            if (__builtin_expect(P.remote_delay != 0, 0)) {           // test hook
                if (mine)
                    __hip_atomic_store(zblock + (size_t)c * ZS + lane, xk, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
                late.hand(P, s_ring, peers_z, lane, mine, xk);
            } else if (mine) {
                peers_z.store(zoff, xk);                             // ... into every replica
            }
        } else if constexpr (!EARLY) {
            if (mine)
                __hip_atomic_store(zblock + (size_t)c * ZS + lane, xk, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);        // publish (sc1)
        }
        // ... and park the bookkeeping columns
        const int slot = (int)(R - R0) & 63;
        lpbuf = (lane == slot) ? -0.5 * cur : lpbuf;
        if (__builtin_expect(slot == 63, 0)) {
            flush_columns(64);
            lp_row += (size_t)64 * n;
            zc_row += (size_t)64 * n;
        }
        pbase = zblock;
        zblock += blockstride;
        MC3_TICK(tk_prevend);
        rdone = R + 1;
    }
    if constexpr (COURIER)                       // (the courier drains the ring and leaves)
        courier_end(s_cour);
    if constexpr (NW > 1)
        helpers.release(lane);
    if constexpr (P2P && !COURIER) {
        if (P.remote_delay != 0)
            late.drain(P, s_ring, peers_z, lane, mine, true);
    }
    flush_columns((int)(rdone - R0) & 63);       // the unwritten tail (also after an early exit)
    if (mine)
        P.X[(size_t)lc * nf + lane] = xk;
    if (lane == 0) {
        P.chisq[lc] = cur;
        P.best_chisq[lc] = best;
        P.numaccept[lc] = nacc0 + nacc;
    }
    MC3_STAMP(4);
#ifdef MC3HIP_STAMPS
    if (lc == 0 && lane == 0) {
        P.stamps[8] += ta8; P.stamps[9] += ta9; P.stamps[10] += ta10;
        P.stamps[11] += ta11; P.stamps[12] += ta12; P.stamps[13] += ta13;
        P.stamps[14] += ta14; P.stamps[15] += ta15;
    }
#endif
}

template <int MT, bool P2P, bool FX = false, bool MULTI = false, bool BATCH = false,
          bool COURIER = false, bool EPS = false, bool BIG = false>
__global__ __launch_bounds__(COURIER ? 128 : 64) void lean_demc_kernel(const DevProblem P,
        const int nref)
{
    lean_demc_body<MT, P2P, FX, MULTI, BATCH, COURIER, EPS, 1, 1, BIG>(P, nref);
}

// ---------------------------------------------------------------------------------
// The snooker sampler (the reference's default) for the same shape, written out by hand: one
// wave per chain, fully free polynomial, thinning = 1, one GPU.  A snooker iteration needs
// three rows of the WHOLE history (chain.py:195-205), almost always published long ago, so
// nothing has to be waited for -- an iteration of the general lean instance still took 1.26 us,
// all of it the instructions between two chisq sums.  Here
//   * the draws of 21 refreshes are computed at once (three Philox slots per refresh, one per
//     lane) TOGETHER with everything that follows from them alone: the row offsets (64-bit
//     bounded draws against the history length of that refresh), the jump flag, gamma_s, the
//     Metropolis draw and the window of its logarithm (handoff.h accept_window); an iteration
//     reads its own with v_readlane;
//   * the rows of the NEXT refresh are asked for before this refresh's chisq is computed;
//   * the ordinary jump (nine in ten) is decided in the log domain; a snooker jump proper
//     (chain.py:207-215, :251-255) takes the exponential and the Hastings factor as ever.
// Same draws, same arithmetic, same results as run_chain's.
// (FX: some coefficients are fixed or shared, or carry Gaussian priors -- as lean_demc_kernel's)
// (THIN: thinning > 1 -- the iterations between two refreshes draw from the same history and
// publish nothing; the draws are keyed by the iteration, R thinning + j, as run_chain's)
// (MULTI: more than 1024 points -- the second tile stays in registers too, the rest is streamed)
// (P2P: the ensemble is sharded over GPUs -- rows are read from the local replica at system
// scope, published into it, and carried to the other ranks' replicas by the COURIER wave, as
// lean_demc_kernel's; without it the chain wave stores them itself)
// (zlag > 0, the declared multi-GPU cadence: the history an iteration draws from ends zlag
// refreshes earlier -- mc3hip_problem.zlag, oracle_lockstep.c ls_run)
template <int MT, bool FX, bool THIN, bool MULTI, bool P2P, bool COURIER, bool EPS = false,
          int NW = 1, int NT = 1>
__device__ __forceinline__ void lean_snooker_body(const DevProblem &P, const int nref)
{
    static_assert(!COURIER || P2P, "the courier carries rows to other ranks' replicas");
    static_assert(NW == 1 || (!P2P && !MULTI && !COURIER), "helper waves: one GPU, one tile");
    constexpr int NC = Model<MT>::NC;         // coefficients the model tag fixes
    const int nf = FX ? P.nfree : NC;
    constexpr int ZS = ZROW;
    // (EPS, fepsilon != 0, chain.py:185,217: two more Philox slots per refresh for the support
    // distribution's deviates of up to four free parameters -- five lanes per refresh; a slot
    // per two coefficients beyond four: the models specialised at run time, up to ten)
    static_assert(!EPS || (!THIN && !MULTI && !P2P), "EPS: one tile, thinning 1, one GPU");
    constexpr int NSE = NC <= 4 ? 2 : (NC + 1) / 2;
    constexpr int LPR = EPS ? 3 + NSE : 3;     // lanes of the batch per refresh
    constexpr int NB = 64 / LPR;               // refreshes per batch of draws
    const int lane = (COURIER || NW > 1) ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const int lc = blockIdx.x;
    const int c = P.chain0 + lc;
    const bool mine = lane < nf;
    const int kk = mine ? lane : 0;
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    __shared__ double s_ring[P2P ? COURIER_RING * ZROW : 1];
    __shared__ int s_cour[4];
    (void)s_ring; (void)s_cour;
    __shared__ double s_hpart[NW > 1 ? 64 * NW : 1], s_hcf[16];
    __shared__ int s_hcmd[2];
    (void)s_hpart; (void)s_hcf; (void)s_hcmd;
    SumHelpers<MT, (NW > 1 ? NW : 2), NT> helpers;     // (gauss.h: helper waves share the data sum)
    (void)helpers;
    if constexpr (COURIER) {
        if (!courier_begin(s_ring, s_cour, ctl_rw))
            return;
        if (threadIdx.x >= 64) {
            courier_run<true>(P, s_ring, s_cour, lane, nf, c);
            return;
        }
    } else if constexpr (NW > 1) {
        if (!helpers.enter(s_hpart, s_hcf, s_hcmd,
                           __hip_atomic_load(&ctl_rw[CTL_ABORT], __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT) != 0))
            return;
    } else {
        if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(
                &ctl_rw[CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
            return;
    }
    // (NT tiles of 1024 points in the registers of the NW waves between them; tiles beyond the
    // data: weight 0)
    DataTile<NW, MT> tiles[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        if (q == 0 || q * VLANES < P.npad) {
            tiles[q].load(P, q * VLANES);
        } else if constexpr (NT > 1) {
            tiles[q].clear();
        }
    }
    const DataTile<NW, MT> &tile = tiles[0];
    if constexpr (NW > 1) {
        if (threadIdx.x >= 64) {
            helpers.serve(P, tiles);
            return;
        }
    }
    int handed = 0;
    (void)handed;
    PeerTable<double> peers_z;
    PeerTableLazy<double> peers_lp;      // (the log_post columns: rare, handoff.h)
    if constexpr (P2P) {
        if constexpr (!COURIER)
            peers_z.load(P.peerZ, P.npeers);
        peers_lp.load(P.peerLP, P.npeers);
    }
    (void)peers_z; (void)peers_lp;
    DelayedRows late;
    late.begin(P, P.ctl[0], c, lane);
    (void)late;
    DataTile<1, MT> tile1;
    if constexpr (MULTI)
        tile1.load(P, VLANES);
    const long long R0 = P.ctl[0];
    const int n = P.nchains;
    const size_t blockstride = (size_t)n * ZS;
    const double *pbase = (R0 == 0) ? P.snap0 : P.Z + (size_t)(P.M0 + (R0 - 1) * n) * ZS;
    double *zblock = P.Z + (size_t)(P.M0 + R0 * n) * ZS;
    double *lp_row = P.log_post + (P.M0 + R0 * n + c);
    int32_t *zc_row = P.zchain + (P.M0 + R0 * n + c);
    double xk = mine ? pbase[(size_t)c * ZS + lane] : 0.0;
    double cur = P.chisq[lc];
    double best = P.best_chisq[lc];
    const long long nacc0 = P.numaccept[lc];
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    const double gamma = P.gamma, feps = P.fepsilon;
    const long long abort_at = (P.test_abort >= 0 && lc == 0) ? (long long)P.test_abort : -1;
    int nacc = 0;
    double lpbuf = 0.0;
    int kf[NC];
    double fx[NC];
    PriorLanes<NC> priors;
    int hkf = 0;                               // (helper waves, FX: what coefficient `lane` is)
    double hfx = 0.0;
    if constexpr (FX) {
        coefficient_map<NC>(P, kf, fx);
        priors.init(P, kf, fx, lane);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            hkf = (lane == j) ? kf[j] : hkf;
            hfx = (lane == j) ? fx[j] : hfx;
        }
    }
    (void)kf; (void)fx; (void)priors; (void)hkf; (void)hfx;
    // lane 3 i + s: slot s of refresh bbase + i.  A, B by slot: 0 -> the offsets (in doubles)
    // of rows iR1, iR2; 1 -> the offset of row iz, the jump flag; 2 -> gamma_s, u (as bits)
    unsigned long long A = 0, B = 0;
    double blo = 0.0, bhi = 0.0;
    const int th = THIN ? P.thinning : 1;
    const long long zlag = P.zlag;
    long long bbase = R0 * th;                 // (the batch counts iterations)
    const int bs = lane % LPR, bi = (lane / LPR < NB) ? lane / LPR : NB - 1;
    double bcs = 0.0, bsn = 0.0;               // EPS, slots 3 and 4: rad cos, rad sin
    const double pstep_k = (EPS && mine) ? P.fstep[lane] : 0.0;
    (void)bcs; (void)bsn; (void)pstep_k;
    auto fill = [&](long long base) {
        const long long Rl = base + bi;
        const Draw2 d = draw2(P.seed, (uint64_t)Rl, (uint32_t)c, (uint32_t)bs);
        // (the history an iteration draws from: the rows before its refresh -- zlag refreshes
        // before it in the declared multi-GPU cadence)
        const long long Rh = (THIN ? Rl / th : Rl) - zlag;
        const unsigned long long zsz = (unsigned long long)(P.M0 + (Rh > 0 ? Rh : 0) * n);
        if (bs == 0) {
            const unsigned long long iR1 = bounded(d.q0, zsz);
            unsigned long long iR2 = 1 + bounded(d.q1, zsz - 1);
            if (iR2 == iR1)
                iR2 = 0;
            A = iR1 * ZS;
            B = iR2 * ZS;
        } else if (bs == 1) {
            A = bounded(d.q1, zsz) * ZS;
            B = (u01(d.q0) < 0.1) ? 1ull : 0ull;
        } else {
            A = mc3_bits(1.2 + (2.2 - 1.2) * u01(d.q0));
            B = mc3_bits(u01(d.q1));
        }
        // (every lane: the lanes of slot 2 keep theirs)
        accept_window(u01(d.q1), blo, bhi);
        if constexpr (EPS) {                   // (every lane: the lanes of slots 3, 4 keep theirs)
            const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(d.q0)));
            double sn, cs;
            mc3_det_sincos2pi(u01(d.q1), &sn, &cs);
            bcs = rad * cs;
            bsn = rad * sn;
        }
        bbase = base;
    };
    auto lane64 = [&](unsigned long long v, int l) {
        return (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l) |
               ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l)
                << 32);
    };
    // what an iteration needs of its draws, and its three rows
    struct Step {
        bool sj;
        double gs, u, lo, hi;
        double normal;                         // EPS: this lane's deviate times its step
        const double *addr[3];
        double val[3];
    };
    auto prepare = [&](long long R, Step &S, bool async) {   // (R inside the current batch)
        const int idx = LPR * (int)(R - bbase);
        S.addr[0] = P.Z + (lane64(A, idx) + (unsigned)kk);
        S.addr[1] = P.Z + (lane64(B, idx) + (unsigned)kk);
        S.addr[2] = P.Z + (lane64(A, idx + 1) + (unsigned)kk);
        if (async) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
                S.val[q] = load_published_async<P2P>(S.addr[q]);
        } else {
#pragma unroll
            for (int q = 0; q < 3; ++q)
                S.val[q] = load_published<P2P>(S.addr[q]);
        }
        S.sj = lane64(B, idx + 1) != 0;
        S.gs = mc3_from_bits(lane64(A, idx + 2));
        S.u = mc3_from_bits(lane64(B, idx + 2));
        S.lo = read_lane(blo, idx + 2);
        S.hi = read_lane(bhi, idx + 2);
        S.normal = 0.0;
        if constexpr (EPS) {
            // lane k's deviate sits in lane idx + 3 + (k >> 1): cosine for even k, sine for odd
            const int src = mine ? idx + 3 + (lane >> 1) : idx;
            const double gc = lane_gather(bcs, src), gn = lane_gather(bsn, src);
            S.normal = mine ? ((lane & 1) ? gn : gc) * pstep_k : 0.0;
        }
    };
    fill(R0 * th);
    Step S;
    prepare(R0 * th, S, true);
    // (retired here: a load the compiler knows of, still pending at the loop head, would make
    // it guard these registers with a wait INSIDE the loop -- for everything in flight)
    published_arrived<0>(S.val[0], S.val[1], S.val[2]);
    // (... and so is everything else loaded so far -- the state, the bounds, the data tile)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const long long Rend = R0 + nref;
    long long rdone = R0;
    auto flush_columns = [&](int count) {
        if (lane < count) {
            if constexpr (P2P) {
                const size_t roff = (size_t)(lp_row - P.log_post) + (size_t)lane * n;
                peers_lp.store(roff, lpbuf);
            } else {
                lp_row[(size_t)lane * n] = lpbuf;
            }
            zc_row[(size_t)lane * n] = c;
        }
    };
    bool alive = true;
    for (long long R = R0; R < Rend; ++R) {
        if (__builtin_expect(R == abort_at, 0)) {                                         // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        // This refresh's rows were asked for at the top of the refresh before and retired at
        // its bottom; the next refresh's are asked for now, first thing, by loads the
        // compiler's wait counts do not see (handoff.h load_published_async): they have the
        // whole iteration to arrive and are retired behind the publish, which they precede
        // in the wave's queue.
        for (int j = 0; j < th; ++j) {
        const long long it = R * th + j;
        Step C = S;
        if (__builtin_expect(it + 1 - bbase == NB, 0))
            fill(it + 1);
        Step Nx;
        prepare(it + 1, Nx, true);
        // (one of the newest rows may still be the sentinel)
        if (__builtin_expect(__any(mc3_bits(C.val[0]) == ROW_PENDING ||
                                   mc3_bits(C.val[1]) == ROW_PENDING ||
                                   mc3_bits(C.val[2]) == ROW_PENDING), 0)) {
            const double *const a3[3] = {C.addr[0], C.addr[1], C.addr[2]};
            double v3[3] = {C.val[0], C.val[1], C.val[2]};
            alive = wait_rows<3, P2P>(a3, v3, ctl_rw, nullptr, [&]() {
                if constexpr (P2P && !COURIER) {             // (test hook: see lean_demc_kernel)
                    if (__builtin_expect(P.remote_delay != 0, 0))
                        late.drain(P, s_ring, peers_z, lane, mine, false);
                }
            });
            // (nothing this path asked for is pending where it joins the other: said in a
            // form the compiler's bookkeeping reads, or it guards the join with a wait of its
            // own -- which the loads of the next refresh, just issued, would be waited for at)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("; rows of this refresh" : "+v"(v3[0]), "+v"(v3[1]), "+v"(v3[2]));
            C.val[0] = v3[0];
            C.val[1] = v3[1];
            C.val[2] = v3[2];
            if (__builtin_expect(!alive, 0))
                goto out;
        }
        const double z1 = C.val[0], z2 = C.val[1], zk = C.val[2];
        const bool sjump = C.sj;
        const double u_now = C.u, lo_now = C.lo, hi_now = C.hi;
        double jump_k;
        if (__builtin_expect(sjump, 0)) {                            // chain.py:207-215
            const double gs = C.gs;
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
        } else {                                                     // chain.py:216-217
            jump_k = gamma * (z1 - z2) + feps * (EPS ? C.normal : 0.0);
        }
        // proposal and boundaries, chain.py:235-243
        const double npk = xk + jump_k;
        const bool outk = mine && (npk < pmin_k || npk > pmax_k);
        if (__builtin_expect(!__any(outk), 1)) {
            if constexpr (NW > 1) {                  // (helper waves: as lean_demc_body's)
                const double got = lane_gather(npk, hkf < 0 ? 0 : hkf);
                helpers.order(FX ? (hkf < 0 ? hfx : got) : npk, lane);
            }
            Model<MT> model;
            if constexpr (!FX) {
                model.init_from_lanes(P, npk);
            } else {
                model.init_from_lanes(P, npk, kf, fx);
            }
            double nextchisq;                                          // chain.py:249
            if constexpr (NW == 1) {
                nextchisq = block_gauss_model<1, MT, MULTI, MULTI>(P, model, nullptr, nullptr,
                                                                   tile, MULTI ? &tile1 : &tile);
            } else {
                double prior = 0.0;
                if constexpr (FX)
                    prior = priors.term(npk, lane);
                nextchisq = helpers.collect(P, model, tiles, lane);
                if constexpr (FX)
                    nextchisq = nextchisq + prior;
            }
            if constexpr (FX && NW == 1)
                nextchisq = nextchisq + priors.term(npk, lane);
            bool acc;
            if (__builtin_expect(!sjump, 1)) {
                acc = metropolis_accept(0.5 * (cur - nextchisq), u_now, lo_now, hi_now);
            } else {                                                 // chain.py:251-255
                const double dc = xk - zk, dn = npk - zk;
                const double cnorm = lane_serial_sum(mine ? dc * dc : 0.0, nf);
                const double nnorm = lane_serial_sum(mine ? dn * dn : 0.0, nf);
                const double mrfactor = mc3_det_pow_half(nnorm / cnorm, nf - 1);
                acc = mc3_det_exp(0.5 * (cur - nextchisq)) * mrfactor > u_now;
            }
            xk = acc ? npk : xk;
            if (__builtin_expect(__any(acc && nextchisq < best), 0)) {
                best = nextchisq;
                if (mine)
                    P.bestx[(size_t)lc * nf + lane] = xk;
                if (lane == 0)
                    P.best_iter[lc] = it;
            }
            cur = acc ? nextchisq : cur;
            nacc += acc ? 1 : 0;
        } else if (outk) {
            atomicAdd(&P.outbounds[lane], 1ull);                     // chain.py:242-243
        }
        // Z append, chain.py:276-289 (THIN: behind the last iteration of the refresh; the
        // others leave the state where the kernel leaves it in the end, so that ONE store
        // follows the loads of the next iteration's rows either way and ONE wait serves both:
        // two waits on two paths for the same registers would have the compiler copy them
        // into place ahead of one of the waits -- a copy of what the loads have not written)
        // (store_published_counted: ONE store instruction, issued on every path, the lanes
        // chosen by EXEC and the destination by a select -- the wait below counts on it; two
        // stores on the two sides of a branch come out of the structuriser as flag-routed
        // blocks with a statically possible path through neither)
        {
            double *const dst = (!THIN || j + 1 == th) ? zblock + (size_t)c * ZS + lane
                                                       : &P.X[(size_t)lc * nf + lane];
            store_published_counted<P2P>(dst, xk, mine);                      // publish
        }
        // the next iteration's rows: everything but the youngest operation of this wave -- the
        // store above at least -- has completed
        published_arrived<1>(Nx.val[0], Nx.val[1], Nx.val[2]);
        S = Nx;
        // ... and to the other ranks' replicas: the courier's business, or -- without one --
        // behind the wait, so that the rows just asked for do not queue behind these stores
        if constexpr (P2P) {
            if (!THIN || j + 1 == th) {
                if constexpr (COURIER) {
                    alive = courier_hand(s_ring, s_cour, handed, lane, mine, xk, ctl_rw);
                    if (__builtin_expect(!alive, 0))
                        goto out;
                } else if (__builtin_expect(P.remote_delay != 0, 0)) {         // test hook
                    late.hand(P, s_ring, peers_z, lane, mine, xk);
                } else if (mine) {
                    peers_z.store((size_t)(zblock - P.Z) + (size_t)c * ZS + lane, xk, P.rank);
                }
            }
        }
        }
        const int slot = (int)(R - R0) & 63;
        lpbuf = (lane == slot) ? -0.5 * cur : lpbuf;
        if (__builtin_expect(slot == 63, 0)) {
            flush_columns(64);
            lp_row += (size_t)64 * n;
            zc_row += (size_t)64 * n;
        }
        zblock += blockstride;
        rdone = R + 1;
    }
out:
    if constexpr (COURIER)
        courier_end(s_cour);
    if constexpr (NW > 1)
        helpers.release(lane);
    if constexpr (P2P && !COURIER) {
        if (P.remote_delay != 0)
            late.drain(P, s_ring, peers_z, lane, mine, true);
    }
    flush_columns((int)(rdone - R0) & 63);
    if (mine)
        P.X[(size_t)lc * nf + lane] = xk;
    if (lane == 0) {
        P.chisq[lc] = cur;
        P.best_chisq[lc] = best;
        P.numaccept[lc] = nacc0 + nacc;
    }
}

template <int MT, bool FX = false, bool THIN = false, bool MULTI = false, bool P2P = false,
          bool COURIER = false, bool EPS = false>
__global__ __launch_bounds__(COURIER ? 128 : 64) void lean_snooker_kernel(const DevProblem P,
        const int nref)
{
    lean_snooker_body<MT, FX, THIN, MULTI, P2P, COURIER, EPS>(P, nref);
}

// 2 049 ... 16 384 points, the plain shape with few chains: the tiles in the registers of four
// (eight) waves per chain -- wave 0 runs the loop above, the others add up their share of every
// tile between two barriers (gauss.h SumHelpers; lean_demc_wide_kernel is the same for DEMC)
template <int MT, int NW, int NT>
__global__ __launch_bounds__(64 * NW) void lean_snooker_wide_kernel(const DevProblem P,
                                                                    const int nref)
{
    lean_snooker_body<MT, false, false, false, false, false, false, NW, NT>(P, nref);
}

// ---------------------------------------------------------------------------------
// The random-walk sampler (chain.py:219-220) for the same shape: no other chain is looked at,
// an iteration is its draws and its chisq -- and the draws (one Philox call for the Metropolis
// draw, one per two parameters for the normal deviates, a logarithm, a square root, a sine
// and a cosine each: chain.py:185) were 0.7 of the 1.2 us the lean run_chain instance took.
// Here they are made for 21 refreshes at once, one (refresh, slot) per lane -- lane 3 i: the
// Metropolis draw and the window of its logarithm; lanes 3 i + 1, 3 i + 2: the deviates of
// parameters 0, 1 and 2, 3 -- and an iteration gathers its own (v_readlane, ds_bpermute).
// (More than four coefficients -- the models specialised at run time, up to ten: a slot of
// deviates per two of them, 16 / 12 / 10 refreshes at once for 5-6 / 7-8 / 9-10 coefficients.)
// Same draws, same arithmetic, same results as run_chain's.
// (THIN: thinning > 1 -- the iterations between two refreshes publish nothing; the draws are
// keyed by the iteration, R thinning + j, as run_chain's)
// (MULTI: more than 1024 points -- the second tile stays in registers too, the rest is streamed)
// (P2P / COURIER: the ensemble is sharded over GPUs; the rows go into every rank's replica, as
// lean_snooker_kernel's -- nobody waits for them: the run needs them complete at its end)
// (NW = 4, NT = 2, 4, 8: more than 1024 points in the registers of four waves, the helper waves
// of gauss.h SumHelpers sharing the data sum -- specialised at run time, round 5)
template <int MT, bool FX, bool THIN, bool MULTI, bool P2P, bool COURIER, int NW = 1, int NT = 1>
__device__ __forceinline__ void lean_mrw_body(const DevProblem &P, const int nref)
{
    static_assert(Model<MT>::NC >= 1 && Model<MT>::NC <= 16, "one lane per coefficient of a row");
    static_assert(!COURIER || P2P, "the courier carries rows to other ranks' replicas");
    static_assert(NW == 1 || (!P2P && !MULTI && !COURIER), "helper waves: one GPU");
    static_assert(NT == 1 || NW > 1, "several tiles: in the registers of several waves");
    constexpr int NC = Model<MT>::NC;         // coefficients the model tag fixes
    const int nf = FX ? P.nfree : NC;
    constexpr int ZS = ZROW;
    // Philox slots of normal deviates per iteration (two deviates each; two slots up to four
    // coefficients, as the instances built with the library have always had: the models
    // specialised at run time have up to ten), lanes of the batch per iteration, iterations
    // per batch
    constexpr int NS = NC <= 4 ? 2 : (NC + 1) / 2;
    constexpr int LPR = NS + 1;
    constexpr int NB = 64 / LPR;
    const int lane = (COURIER || NW > 1) ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const int lc = blockIdx.x;
    const int c = P.chain0 + lc;
    const bool mine = lane < nf;
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    __shared__ double s_ring[P2P ? COURIER_RING * ZROW : 1];
    __shared__ int s_cour[4];
    (void)s_ring; (void)s_cour;
    __shared__ double s_hpart[NW > 1 ? 64 * NW : 1], s_hcf[16];
    __shared__ int s_hcmd[2];
    (void)s_hpart; (void)s_hcf; (void)s_hcmd;
    SumHelpers<MT, (NW > 1 ? NW : 2), NT> helpers;     // (gauss.h: helper waves share the data sum)
    (void)helpers;
    if constexpr (COURIER) {
        if (!courier_begin(s_ring, s_cour, ctl_rw))
            return;
        if (threadIdx.x >= 64) {
            courier_run<true>(P, s_ring, s_cour, lane, nf, c);
            return;
        }
    } else if constexpr (NW > 1) {
        if (!helpers.enter(s_hpart, s_hcf, s_hcmd,
                           __hip_atomic_load(&ctl_rw[CTL_ABORT], __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT) != 0))
            return;
    } else {
        if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(
                &ctl_rw[CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
            return;
    }
    int handed = 0;
    (void)handed;
    PeerTable<double> peers_z;
    PeerTableLazy<double> peers_lp;      // (the log_post columns: rare, handoff.h)
    if constexpr (P2P) {
        if constexpr (!COURIER)
            peers_z.load(P.peerZ, P.npeers);
        peers_lp.load(P.peerLP, P.npeers);
    }
    (void)peers_z; (void)peers_lp;
    DelayedRows late;
    late.begin(P, P.ctl[0], c, lane);
    (void)late;
    // (NT tiles of 1024 points in the registers of the NW waves between them; tiles beyond the
    // data: weight 0)
    DataTile<NW, MT> tiles[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        if (q == 0 || q * VLANES < P.npad) {
            tiles[q].load(P, q * VLANES);
        } else if constexpr (NT > 1) {
            tiles[q].clear();
        }
    }
    const DataTile<NW, MT> &tile = tiles[0];
    if constexpr (NW > 1) {
        if (threadIdx.x >= 64) {
            helpers.serve(P, tiles);
            return;
        }
    }
    DataTile<1, MT> tile1;
    if constexpr (MULTI)
        tile1.load(P, VLANES);
    const long long R0 = P.ctl[0];
    const int n = P.nchains;
    const size_t blockstride = (size_t)n * ZS;
    const double *pbase = (R0 == 0) ? P.snap0 : P.Z + (size_t)(P.M0 + (R0 - 1) * n) * ZS;
    double *zblock = P.Z + (size_t)(P.M0 + R0 * n) * ZS;
    double *lp_row = P.log_post + (P.M0 + R0 * n + c);
    int32_t *zc_row = P.zchain + (P.M0 + R0 * n + c);
    double xk = mine ? pbase[(size_t)c * ZS + lane] : 0.0;
    double cur = P.chisq[lc];
    double best = P.best_chisq[lc];
    const long long nacc0 = P.numaccept[lc];
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    const double pstep_k = mine ? P.fstep[lane] : 0.0;
    const long long abort_at = (P.test_abort >= 0 && lc == 0) ? (long long)P.test_abort : -1;
    int nacc = 0;
    double lpbuf = 0.0;
    int kf[NC];
    double fx[NC];
    PriorLanes<NC> priors;
    int hkf = 0;                               // (helper waves, FX: what coefficient `lane` is)
    double hfx = 0.0;
    if constexpr (FX) {
        coefficient_map<NC>(P, kf, fx);
        priors.init(P, kf, fx, lane);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            hkf = (lane == j) ? kf[j] : hkf;
            hfx = (lane == j) ? fx[j] : hfx;
        }
    }
    (void)kf; (void)fx; (void)priors; (void)hkf; (void)hfx;
    // lane LPR i + s of the batch: s = 0 -> u (bu) and its window; s = 1 .. NS -> rad cos, rad
    // sin of Philox slot 3 + (s - 1), i.e. the deviates of parameters 2 (s - 1) and 2 (s - 1) + 1
    double bu = 0.0, blo = 0.0, bhi = 0.0, bcs = 0.0, bsn = 0.0;
    const int th = THIN ? P.thinning : 1;
    long long bbase = R0 * th;                 // (the batch counts iterations)
    const int bs = lane % LPR, bi = (lane / LPR < NB) ? lane / LPR : NB - 1;
    auto fill = [&](long long base) {
        const Draw2 d = draw2(P.seed, (uint64_t)(base + bi), (uint32_t)c,
                              bs == 0 ? 0u : (uint32_t)(2 + bs));
        bu = u01(d.q1);                                              // chain.py:259
        accept_window(bu, blo, bhi);
        const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(d.q0)));
        double sn, cs;
        mc3_det_sincos2pi(u01(d.q1), &sn, &cs);
        bcs = rad * cs;
        bsn = rad * sn;
        bbase = base;
    };
    fill(R0 * th);
    // lane k's deviate of a refresh sits in lane LPR i + 1 + (k >> 1): cosine branch for even k
    const int src_off = 1 + (lane >> 1);
    const long long Rend = R0 + nref;
    long long rdone = R0;
    auto flush_columns = [&](int count) {
        if (lane < count) {
            if constexpr (P2P) {
                const size_t roff = (size_t)(lp_row - P.log_post) + (size_t)lane * n;
                peers_lp.store(roff, lpbuf);
            } else {
                lp_row[(size_t)lane * n] = lpbuf;
            }
            zc_row[(size_t)lane * n] = c;
        }
    };
    bool alive = true;
    // (everything loaded so far -- the state, the bounds, the data tile -- retired here: a load
    // still pending at the loop head would make the compiler guard its registers with a wait
    // INSIDE the loop, i.e. for the publish of the iteration before as well)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (long long R = R0; R < Rend; ++R) {
        if (__builtin_expect(R == abort_at, 0)) {                                         // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        for (int j = 0; j < th; ++j) {
        const long long it = R * th + j;
        if (__builtin_expect(it - bbase == NB, 0))
            fill(it);
        const int idx = LPR * (int)(it - bbase);
        const double u_now = read_lane(bu, idx), lo_now = read_lane(blo, idx),
                     hi_now = read_lane(bhi, idx);
        const int src = mine ? idx + src_off : idx;
        const double gc = lane_gather(bcs, src), gs = lane_gather(bsn, src);
        // (rad * cos or sin) * pstep, as run_chain's normal_draw
        const double jump_k = mine ? ((lane & 1) ? gs : gc) * pstep_k : 0.0;
        // proposal and boundaries, chain.py:235-243
        const double npk = xk + jump_k;
        const bool outk = mine && (npk < pmin_k || npk > pmax_k);
        if (__builtin_expect(!__any(outk), 1)) {
            // (helper waves: woken first, as in lean_demc_body)
            if constexpr (NW > 1) {
                const double got = lane_gather(npk, hkf < 0 ? 0 : hkf);
                helpers.order(FX ? (hkf < 0 ? hfx : got) : npk, lane);
            }
            Model<MT> model;
            if constexpr (!FX) {
                model.init_from_lanes(P, npk);
            } else {
                model.init_from_lanes(P, npk, kf, fx);
            }
            double nextchisq;                                        // chain.py:249
            if constexpr (NW == 1) {
                nextchisq = block_gauss_model<1, MT, MULTI, MULTI>(P, model, nullptr, nullptr,
                                                                   tile, MULTI ? &tile1 : &tile);
                if constexpr (FX)
                    nextchisq = nextchisq + priors.term(npk, lane);
            } else {
                double prior = 0.0;
                if constexpr (FX)
                    prior = priors.term(npk, lane);
                nextchisq = helpers.collect(P, model, tiles, lane);
                if constexpr (FX)
                    nextchisq = nextchisq + prior;
            }
            const bool acc = metropolis_accept(0.5 * (cur - nextchisq), u_now, lo_now, hi_now);
            xk = acc ? npk : xk;
            if (__builtin_expect(__any(acc && nextchisq < best), 0)) {
                best = nextchisq;
                if (mine)
                    P.bestx[(size_t)lc * nf + lane] = xk;
                if (lane == 0)
                    P.best_iter[lc] = it;
            }
            cur = acc ? nextchisq : cur;
            nacc += acc ? 1 : 0;
        } else if (outk) {
            atomicAdd(&P.outbounds[lane], 1ull);                     // chain.py:242-243
        }
        }
        // Z append, chain.py:276-289
        if (mine)
            __hip_atomic_store(zblock + (size_t)c * ZS + lane, xk, __ATOMIC_RELAXED,
                               P2P ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (P2P) {
            if constexpr (COURIER) {
                alive = courier_hand(s_ring, s_cour, handed, lane, mine, xk, ctl_rw);
                if (__builtin_expect(!alive, 0))
                    break;
            } else if (__builtin_expect(P.remote_delay != 0, 0)) {             // test hook
                late.hand(P, s_ring, peers_z, lane, mine, xk);
            } else if (mine) {
                peers_z.store((size_t)(zblock - P.Z) + (size_t)c * ZS + lane, xk, P.rank);
            }
        }
        const int slot = (int)(R - R0) & 63;
        lpbuf = (lane == slot) ? -0.5 * cur : lpbuf;
        if (__builtin_expect(slot == 63, 0)) {
            flush_columns(64);
            lp_row += (size_t)64 * n;
            zc_row += (size_t)64 * n;
        }
        zblock += blockstride;
        rdone = R + 1;
    }
    if constexpr (COURIER)
        courier_end(s_cour);
    if constexpr (NW > 1)
        helpers.release(lane);
    if constexpr (P2P && !COURIER) {
        if (P.remote_delay != 0)
            late.drain(P, s_ring, peers_z, lane, mine, true);
    }
    (void)alive;
    flush_columns((int)(rdone - R0) & 63);
    if (mine)
        P.X[(size_t)lc * nf + lane] = xk;
    if (lane == 0) {
        P.chisq[lc] = cur;
        P.best_chisq[lc] = best;
        P.numaccept[lc] = nacc0 + nacc;
    }
}

template <int MT, bool FX = false, bool THIN = false, bool MULTI = false, bool P2P = false,
          bool COURIER = false>
__global__ __launch_bounds__(COURIER ? 128 : 64) void lean_mrw_kernel(const DevProblem P,
        const int nref)
{
    lean_mrw_body<MT, FX, THIN, MULTI, P2P, COURIER>(P, nref);
}

// ---------------------------------------------------------------------------------
// DEMC in the DECLARED MULTI-GPU CADENCE (mc3hip_problem.zlag = D >= 1): during refresh R a chain
// sees the others through the rows of refresh R - 1 - D (the initial snapshot until there is
// one) -- oracle_lockstep.c ls_run states it.  In the strict lockstep an iteration is
// "wait for the partners' rows of the iteration before, then compute": 0.53 us of hand-off on
// one chip in front of 0.63 us of arithmetic, and h_x in front of it across xGMI (DESIGN.md
// section 7).  With D >= 1 the rows an iteration needs were published D iterations ago: they
// have D iterations' time to cross the fabric, and an iteration costs max(arithmetic, h_x / D).
// Nothing is waited for in the common case, so the loop is laid out like lean_snooker_kernel's:
// the draws of 64 iterations at once, one per lane (Philox, log u and its window, the partners'
// row offsets); the two rows of the NEXT iteration asked for first
// thing by loads the compiler does not see (handoff.h load_published_async: one load site, one
// wait site) and retired behind this iteration's store; a row that is still the sentinel --
// D x the iteration time shorter than the hop -- is asked for again the usual way.
//   any thinning (the `thinning` iterations of a refresh draw their partners anew but look at
//   the same rows) -- also in the STRICT lockstep, zlag = 0 with thinning > 1, where only the
//   first iteration of a refresh asks for rows its partners are about to publish and waits
//   (round 3's lean_demc_thin_kernel was this loop for the plain shape: 290 -> 359 M
//   chain-steps/s at thinning 10);
//   FX: some coefficients fixed or shared, Gaussian priors (as lean_demc_kernel's);
//   P2P: the ensemble is sharded over GPUs -- rows are read from the local replica at system
//   scope, published into it, and carried to the other replicas by the COURIER wave (handoff.h;
//   without it the chain wave stores them itself: more than two ranks sharing a device).
// Same draws, same arithmetic, same results as run_chain's with the same zlag -- on one GPU or
// sharded over any number of them.
// (Asking TWO refreshes ahead -- the loop is bound by the round trip of its own loads, 0.84 us
// per iteration against the random walk's 0.58 -- was tried in round 4 with the rows landing in
// fixed accumulation registers, two sets used in turn by a loop unrolled by two: bit-identical,
// SLOWER (213 against 306 M chain-steps/s at 256 chains) and over the registers of two waves per
// SIMD; not kept.)
// (NW = 4 / 8, NT = 2 ... 16: more than 1024 points in the registers of several waves, the helper
// waves of gauss.h SumHelpers sharing the data sum -- thinning > 1 beyond one tile, specialised
// at run time, round 5)
template <int MT, bool P2P, bool FX, bool COURIER, int NW = 1, int NT = 1>
__device__ __forceinline__ void lean_demc_lag_body(const DevProblem &P, const int nref)
{
    static_assert(!COURIER || P2P, "the courier carries rows to other ranks' replicas");
    static_assert(NW == 1 || (!P2P && !COURIER), "helper waves: one GPU");
    static_assert(NT == 1 || NW > 1, "several tiles: in the registers of several waves");
    constexpr int ZS = ZROW;
    constexpr int NB = 64;                     // iterations per batch of draws (one per lane)
    constexpr int NC = Model<MT>::NC;         // coefficients the model tag fixes
    const int nf = FX ? P.nfree : NC;
    const int lane = (COURIER || NW > 1) ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const int lc = blockIdx.x;
    const int c = P.chain0 + lc;
    const bool mine = lane < nf;
    const int kk = mine ? lane : 0;
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    __shared__ double s_ring[P2P ? COURIER_RING * ZROW : 1];
    __shared__ int s_cour[4];
    (void)s_ring; (void)s_cour;
    __shared__ double s_hpart[NW > 1 ? 64 * NW : 1], s_hcf[16];
    __shared__ int s_hcmd[2];
    (void)s_hpart; (void)s_hcf; (void)s_hcmd;
    SumHelpers<MT, (NW > 1 ? NW : 2), NT> helpers;     // (gauss.h: helper waves share the data sum)
    (void)helpers;
    if constexpr (COURIER) {
        if (!courier_begin(s_ring, s_cour, ctl_rw))
            return;
        if (threadIdx.x >= 64) {
            courier_run<true>(P, s_ring, s_cour, lane, nf, c);
            return;
        }
    } else if constexpr (NW > 1) {
        if (!helpers.enter(s_hpart, s_hcf, s_hcmd,
                           __hip_atomic_load(&ctl_rw[CTL_ABORT], __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT) != 0))
            return;
    } else {
        if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(
                &ctl_rw[CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
            return;
    }
    int handed = 0;
    (void)handed;
    PeerTable<double> peers_z;
    PeerTableLazy<double> peers_lp;      // (the log_post columns: rare, handoff.h)
    if constexpr (P2P) {
        if constexpr (!COURIER)
            peers_z.load(P.peerZ, P.npeers);
        peers_lp.load(P.peerLP, P.npeers);
    }
    (void)peers_z; (void)peers_lp;
    DelayedRows late;
    late.begin(P, P.ctl[0], c, lane);
    (void)late;
    // (NT tiles of 1024 points in the registers of the NW waves between them; tiles beyond the
    // data: weight 0)
    DataTile<NW, MT> tiles[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        if (q == 0 || q * VLANES < P.npad)
            tiles[q].load(P, q * VLANES);
        else if constexpr (NT > 1)
            tiles[q].clear();
    }
    const DataTile<NW, MT> &tile = tiles[0];
    if constexpr (NW > 1) {
        if (threadIdx.x >= 64) {
            helpers.serve(P, tiles);
            return;
        }
    }
    const long long R0 = P.ctl[0];
    const int n = P.nchains;
    const int th = P.thinning;
    const long long D = P.zlag;
    const size_t blockstride = (size_t)n * ZS;
    // the rows refresh R looks at: of refresh R - 1 - D
    auto seen_by = [&](long long R) -> const double * {
        const long long Rs = R - 1 - D;
        return (Rs < 0) ? P.snap0 : P.Z + (size_t)(P.M0 + Rs * n) * ZS;
    };
    const double *own = (R0 == 0) ? P.snap0 : P.Z + (size_t)(P.M0 + (R0 - 1) * n) * ZS;
    const double *sb = seen_by(R0);
    double *zblock = P.Z + (size_t)(P.M0 + R0 * n) * ZS;
    double *lp_row = P.log_post + (P.M0 + R0 * n + c);
    int32_t *zc_row = P.zchain + (P.M0 + R0 * n + c);
    double xk = mine ? own[(size_t)c * ZS + lane] : 0.0;
    double cur = P.chisq[lc];
    double best = P.best_chisq[lc];
    const long long nacc0 = P.numaccept[lc];
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    const double gamma = P.gamma, feps = P.fepsilon;
    const long long abort_at = (P.test_abort >= 0 && lc == 0) ? (long long)P.test_abort : -1;
    long long nacc = 0;
    double lpbuf = 0.0;
    int kf[NC];
    double fx[NC];
    PriorLanes<NC> priors;
    int hkf = 0;                               // (helper waves, FX: what coefficient `lane` is)
    double hfx = 0.0;
    if constexpr (FX) {
        coefficient_map<NC>(P, kf, fx);
        priors.init(P, kf, fx, lane);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            hkf = (lane == j) ? kf[j] : hkf;
            hfx = (lane == j) ? fx[j] : hfx;
        }
    }
    (void)kf; (void)fx; (void)priors; (void)hkf; (void)hfx;
    // lane i: the draws of iteration bbase + i (chain.py:221-232, :259)
    double bu = 0.0, blo = 0.0, bhi = 0.0;
    int bo1 = 0, bo2 = 0;
    long long bbase = R0 * th;
    auto fill = [&](long long base) {
        const Draw2 dd = draw2(P.seed, (uint64_t)(base + lane), (uint32_t)c, 0u);
        bu = u01(dd.q1);
        accept_window(bu, blo, bhi);
        int q1, q2;
        demc_partners(dd.q0, n, c, q1, q2);
        bo1 = q1 * ZS;
        bo2 = q2 * ZS;
        bbase = base;
    };
    struct Step {
        double u, lo, hi;
        const double *addr[2];
        double val[2];
    };
    // (`it` inside the current batch; `snap`: the rows its refresh looks at)
    auto prepare = [&](long long it, const double *snap, Step &S) {
        const int idx = (int)(it - bbase);
        S.addr[0] = snap + (__builtin_amdgcn_readlane(bo1, idx) + kk);
        S.addr[1] = snap + (__builtin_amdgcn_readlane(bo2, idx) + kk);
        S.val[0] = load_published_async<P2P>(S.addr[0]);
        S.val[1] = load_published_async<P2P>(S.addr[1]);
        S.u = read_lane(bu, idx);
        S.lo = read_lane(blo, idx);
        S.hi = read_lane(bhi, idx);
    };
    fill(R0 * th);
    Step S;
    double spare = 0.0;                        // (published_arrived takes three)
    prepare(R0 * th, sb, S);
    published_arrived<0>(S.val[0], S.val[1], spare);
    // (... and so is everything else loaded so far: a load the compiler knows of, still pending
    // at the loop head, would make it guard its registers with a wait INSIDE the loop)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const long long Rend = R0 + nref;
    long long rdone = R0;
    auto flush_columns = [&](int count) {
        if (lane < count) {
            if constexpr (P2P) {
                const size_t roff = (size_t)(lp_row - P.log_post) + (size_t)lane * n;
                peers_lp.store(roff, lpbuf);
            } else {
                lp_row[(size_t)lane * n] = lpbuf;
            }
            zc_row[(size_t)lane * n] = c;
        }
    };
    bool alive = true;
    for (long long R = R0; R < Rend; ++R) {
        if (__builtin_expect(R == abort_at, 0)) {                                         // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        const double *const sb_next = seen_by(R + 1);
        for (int j = 0; j < th; ++j) {
            const long long it = R * th + j;
            Step C = S;
            if (__builtin_expect(it + 1 - bbase == NB, 0))
                fill(it + 1);
            // the next iteration's rows: of the same refresh, or -- behind the last iteration of
            // this refresh -- those the next refresh looks at, published D refreshes ago
            Step Nx;
            prepare(it + 1, (j + 1 < th) ? sb : sb_next, Nx);
            if (__builtin_expect(__any(mc3_bits(C.val[0]) == ROW_PENDING ||
                                       mc3_bits(C.val[1]) == ROW_PENDING), 0)) {
                const double *const a2[2] = {C.addr[0], C.addr[1]};
                double v2[2] = {C.val[0], C.val[1]};
                alive = wait_rows<2, P2P>(a2, v2, ctl_rw, nullptr, [&]() {
                    if constexpr (P2P && !COURIER) {         // (test hook: see lean_demc_kernel)
                        if (__builtin_expect(P.remote_delay != 0, 0))
                            late.drain(P, s_ring, peers_z, lane, mine, false);
                    }
                });
                // (nothing this path asked for is pending where it joins the other)
                __builtin_amdgcn_s_waitcnt(0x0F70);
                asm volatile("; rows of this iteration" : "+v"(v2[0]), "+v"(v2[1]));
                C.val[0] = v2[0];
                C.val[1] = v2[1];
                if (__builtin_expect(!alive, 0))
                    goto out;
            }
            // proposal and boundaries, chain.py:232-243
            const double jump_k = gamma * (C.val[0] - C.val[1]) + feps * 0.0;
            const double npk = xk + jump_k;
            const bool outk = mine && (npk < pmin_k || npk > pmax_k);
            if (__builtin_expect(!__any(outk), 1)) {
                if constexpr (NW > 1) {                // (helper waves: woken first)
                    const double got = lane_gather(npk, hkf < 0 ? 0 : hkf);
                    helpers.order(FX ? (hkf < 0 ? hfx : got) : npk, lane);
                }
                Model<MT> model;
                if constexpr (!FX)
                    model.init_from_lanes(P, npk);
                else
                    model.init_from_lanes(P, npk, kf, fx);
                double nextchisq;                                        // chain.py:249
                if constexpr (NW == 1) {
                    nextchisq = block_gauss_model<1, MT, false, false>(P, model, nullptr, nullptr,
                                                                       tile, &tile);
                    if constexpr (FX)
                        nextchisq = nextchisq + priors.term(npk, lane);
                } else {
                    double prior = 0.0;
                    if constexpr (FX)
                        prior = priors.term(npk, lane);
                    nextchisq = helpers.collect(P, model, tiles, lane);
                    if constexpr (FX)
                        nextchisq = nextchisq + prior;
                }
                const bool acc = metropolis_accept(0.5 * (cur - nextchisq), C.u, C.lo, C.hi);
                xk = acc ? npk : xk;
                if (__builtin_expect(__any(acc && nextchisq < best), 0)) {
                    best = nextchisq;
                    if (mine)
                        P.bestx[(size_t)lc * nf + lane] = xk;
                    if (lane == 0)
                        P.best_iter[lc] = it;
                }
                cur = acc ? nextchisq : cur;
                nacc += acc ? 1 : 0;
            } else if (outk) {
                atomicAdd(&P.outbounds[lane], 1ull);                     // chain.py:242-243
            }
            // Z append behind the last iteration of the refresh, chain.py:276-289; the others
            // leave the state where the kernel leaves it in the end, so that ONE store follows
            // the loads of the next iteration's rows either way and one wait serves both
            // (store_published_counted: ONE store instruction on every path, the destination
            // chosen by a select -- the wait below counts on it)
            {
                double *const dst = (j + 1 == th) ? zblock + (size_t)c * ZS + lane
                                                  : &P.X[(size_t)lc * nf + lane];
                store_published_counted<P2P>(dst, xk, mine);                  // publish
            }
            published_arrived<1>(Nx.val[0], Nx.val[1], spare);
            S = Nx;
            // ... and to the other ranks' replicas: the courier's business, or -- without one --
            // behind the wait, so that the rows just asked for do not queue behind these stores
            if constexpr (P2P) {
                if (j + 1 == th) {
                    if constexpr (COURIER) {
                        alive = courier_hand(s_ring, s_cour, handed, lane, mine, xk, ctl_rw);
                        if (__builtin_expect(!alive, 0))
                            goto out;
                    } else if (__builtin_expect(P.remote_delay != 0, 0)) {     // test hook
                        late.hand(P, s_ring, peers_z, lane, mine, xk);
                    } else if (mine) {
                        peers_z.store((size_t)(zblock - P.Z) + (size_t)c * ZS + lane, xk, P.rank);
                    }
                }
            }
        }
        {
            const int slot = (int)(R - R0) & 63;
            lpbuf = (lane == slot) ? -0.5 * cur : lpbuf;
            if (__builtin_expect(slot == 63, 0)) {
                flush_columns(64);
                lp_row += (size_t)64 * n;
                zc_row += (size_t)64 * n;
            }
        }
        sb = sb_next;
        zblock += blockstride;
        rdone = R + 1;
    }
out:
    if constexpr (COURIER)
        courier_end(s_cour);
    if constexpr (NW > 1)
        helpers.release(lane);
    if constexpr (P2P && !COURIER) {
        if (P.remote_delay != 0)
            late.drain(P, s_ring, peers_z, lane, mine, true);
    }
    flush_columns((int)(rdone - R0) & 63);
    if (mine)
        P.X[(size_t)lc * nf + lane] = xk;
    if (lane == 0) {
        P.chisq[lc] = cur;
        P.best_chisq[lc] = best;
        P.numaccept[lc] = nacc0 + nacc;
    }
}

template <int MT, bool P2P = false, bool FX = false, bool COURIER = false>
__global__ __launch_bounds__(COURIER ? 128 : 64) void lean_demc_lag_kernel(const DevProblem P,
        const int nref)
{
    lean_demc_lag_body<MT, P2P, FX, COURIER>(P, nref);
}

// ---------------------------------------------------------------------------------
// The same loop for ensembles with more chains than waves can be co-resident (P.mc_chains = T,
// 2..3; one GPU) whose data fill TWO tiles (MULTI; the one-tile shapes of every sampler have the
// chains of a wave side by side instead: kernels_rows.h): a wave owns chains b, b + G, ... b + (T - 1) G of a grid of G and advances
// them one after another inside every refresh -- the turns of run_chain MC.  There a turn
// costs ~1.1 us whatever the occupancy (tools/mc_probe.py), of which the arithmetic is ~0.5:
// every turn waits for the round trip of its own partner rows.  Here
//   * the partner rows of ALL turns of a refresh are asked for at its top -- they are rows of
//     the refresh before whichever turn needs them -- so one round trip per refresh is
//     exposed, not one per turn; nothing is stored before they have all arrived (a wait behind
//     a store waits for the store's acknowledgement as well: loads and stores share vmcnt);
//   * T is a compile-time constant and the chains' states stay in registers: no LDS;
//   * the draws are computed for 64 / T refreshes x T turns at once, one (refresh, turn) per
//     lane, as in lean_demc_kernel BATCH.
// The hand-off protocol is unchanged (a chain waits only for rows of the refresh before, and a
// wave finishes refresh R - 1 for all its chains before it starts R); same arithmetic, same
// results as run_chain's.
template <int MT, int T, bool FX = false, bool MULTI = false>
__global__ __launch_bounds__(64) void lean_demc_mc_kernel(const DevProblem P, const int nref)
{
    static_assert(T >= 2 && T <= 4, "2..4 chains per wave");
    const int nf = FX ? P.nfree : MT;
    constexpr int ZS = ZROW;
    constexpr int NB = 64 / T;                 // refreshes per batch of draws
    const int lane = threadIdx.x;
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const bool mine = lane < nf;
    const int kk = mine ? lane : 0;
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(
            &ctl_rw[CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
        return;
    DataTile<1, MT> tile;
    tile.load(P);
    DataTile<1, MT> tile1;
    if constexpr (MULTI)
        tile1.load(P, VLANES);
    const long long R0 = P.ctl[0];
    const int n = P.nchains;
    const size_t blockstride = (size_t)n * ZS;
    const double *pbase = (R0 == 0) ? P.snap0 : P.Z + (size_t)(P.M0 + (R0 - 1) * n) * ZS;
    double *zblock = P.Z + (size_t)(P.M0 + R0 * n) * ZS;
    double *lp_row = P.log_post + (P.M0 + R0 * n);
    int32_t *zc_row = P.zchain + (P.M0 + R0 * n);
    // this wave's chains (the last may not exist: its turn is skipped)
    bool have[T];
    int lq[T];
    double xk[T], cur[T], best[T];
    int nacc[T];
#pragma unroll
    for (int q = 0; q < T; ++q) {
        have[q] = b + q * G < P.nlocal;
        lq[q] = have[q] ? b + q * G : b;
        xk[q] = mine ? pbase[(size_t)(P.chain0 + lq[q]) * ZS + lane] : 0.0;
        cur[q] = P.chisq[lq[q]];
        best[q] = P.best_chisq[lq[q]];
        nacc[q] = 0;
    }
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    const double gamma = P.gamma, feps = P.fepsilon;
    const long long abort_at = (P.test_abort >= 0 && b == 0) ? (long long)P.test_abort : -1;
    bool alive = true;
    int kf[MT];
    double fx[MT];
    PriorLanes<MT> priors;
    if constexpr (FX) {
        coefficient_map<MT>(P, kf, fx);
        priors.init(P, kf, fx, lane);
    }
    (void)kf; (void)fx; (void)priors;
    // the draws: lane r T + q holds those of refresh bbase + r, turn q (chain.py:221-232, :259)
    const int bq = lane % T, br = (lane / T < NB) ? lane / T : NB - 1;
    const int bl = (b + bq * G < P.nlocal) ? b + bq * G : b;
    const int bc = P.chain0 + bl;
    double bu, blo, bhi;
    int bo1, bo2;
    long long bbase;
    auto fill = [&](long long base) {
        const Draw2 dd = draw2(P.seed, (uint64_t)(base + br), (uint32_t)bc, 0u);
        bu = u01(dd.q1);
        accept_window(bu, blo, bhi);
        int r1, r2;
        demc_partners(dd.q0, n, bc, r1, r2);
        bo1 = r1 * ZS;
        bo2 = r2 * ZS;
        bbase = base;
    };
    fill(R0);
    const long long Rend = R0 + nref;
    // (everything loaded so far retired here: a load still pending at the loop head would make
    // the compiler guard its registers with a wait INSIDE the loop -- for the stores of the
    // turn before as well)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (long long R = R0; R < Rend; ++R) {
        if (__builtin_expect(R == abort_at, 0)) {                                         // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            break;                // (the host restores its checkpoint after an abort)
        }
        __builtin_amdgcn_s_sleep(1);
        // the partner rows of every turn of this refresh, asked for at once
        const int idx0 = (int)(R - bbase) * T;
        const double *addr[T][2];
        double val[T][2], u_q[T], lo_q[T], hi_q[T];
#pragma unroll
        for (int q = 0; q < T; ++q) {
            addr[q][0] = pbase + (__builtin_amdgcn_readlane(bo1, idx0 + q) + kk);
            addr[q][1] = pbase + (__builtin_amdgcn_readlane(bo2, idx0 + q) + kk);
            val[q][0] = load_published<false>(addr[q][0]);
            val[q][1] = load_published<false>(addr[q][1]);
            u_q[q] = read_lane(bu, idx0 + q);
            lo_q[q] = read_lane(blo, idx0 + q);
            hi_q[q] = read_lane(bhi, idx0 + q);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (__builtin_expect(R - bbase == NB - 1, 0)) {
            // ... and behind them, once per batch, the draws of the next NB refreshes
            fill(R + 1);
            asm volatile("; pin" : "+v"(bu), "+v"(blo), "+v"(bhi), "+v"(bo1), "+v"(bo2));
        }
        __builtin_amdgcn_sched_barrier(0);
        // The rows of ALL turns are waited for here (asked for back to back, they arrive within
        // a few cycles of each other): a wait further down, behind the stores of a turn, would
        // wait for those stores' acknowledgements too (loads and stores share vmcnt) -- and the
        // compiler puts one at every use of a register it believes a load may still be on its
        // way to, however the path got there.
        __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
        for (int q = 0; q < T; ++q) {
            if (!have[q])
                continue;
            const int l = lq[q];
            const int c = P.chain0 + l;
            // what is still the sentinel is asked for again (the rows of the refresh before are
            // the last the other waves have published)
            if (__builtin_expect(__any(mc3_bits(val[q][0]) == ROW_PENDING ||
                                       mc3_bits(val[q][1]) == ROW_PENDING), 0)) {
                const double *const a2[2] = {addr[q][0], addr[q][1]};
                double v2[2] = {val[q][0], val[q][1]};
                alive = wait_rows<2, false>(a2, v2, ctl_rw);
                // (nothing this path asked for is pending where it joins the other: said in a
                // form the compiler's bookkeeping reads, or it guards the join -- and the loop
                // head -- with a wait of its own, which the publish of the turn before, just
                // issued, would be waited for at)
                __builtin_amdgcn_s_waitcnt(0x0F70);
                asm volatile("; rows of this turn" : "+v"(v2[0]), "+v"(v2[1]));
                val[q][0] = v2[0];
                val[q][1] = v2[1];
                if (__builtin_expect(!alive, 0))
                    break;
            }
            // proposal and boundaries, chain.py:232-243
            const double jump_k = gamma * (val[q][0] - val[q][1]) + feps * 0.0;
            const double npk = xk[q] + jump_k;
            const bool outk = mine && (npk < pmin_k || npk > pmax_k);
            if (__builtin_expect(!__any(outk), 1)) {
                Model<MT> model;
                if constexpr (!FX) {
                    model.init_from_lanes(P, npk);
                } else {
                    model.init_from_lanes(P, npk, kf, fx);
                }
                double nextchisq =
                    block_gauss_model<1, MT, MULTI, MULTI>(P, model, nullptr, nullptr, tile,
                                                           &tile1);          // chain.py:249
                if constexpr (FX)
                    nextchisq = nextchisq + priors.term(npk, lane);
                const bool acc = metropolis_accept(0.5 * (cur[q] - nextchisq), u_q[q], lo_q[q],
                                                   hi_q[q]);            // chain.py:257-274
                xk[q] = acc ? npk : xk[q];
                if (__builtin_expect(__any(acc && nextchisq < best[q]), 0)) {
                    best[q] = nextchisq;
                    if (mine)
                        P.bestx[(size_t)l * nf + lane] = xk[q];
                    if (lane == 0)
                        P.best_iter[l] = R;
                }
                cur[q] = acc ? nextchisq : cur[q];
                nacc[q] += acc ? 1 : 0;
            } else if (outk) {
                atomicAdd(&P.outbounds[lane], 1ull);                     // chain.py:242-243
            }
            // Z append, chain.py:276-289
            if (mine)
                __hip_atomic_store(zblock + (size_t)c * ZS + lane, xk[q], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);            // publish (sc1)
            if (lane == 0) {
                lp_row[c] = -0.5 * cur[q];
                zc_row[c] = c;
            }
        }
        if (__builtin_expect(!alive, 0))
            break;
        pbase = zblock;
        zblock += blockstride;
        lp_row += n;
        zc_row += n;
    }
#pragma unroll
    for (int q = 0; q < T; ++q) {
        if (!have[q])
            continue;
        const int l = lq[q];
        if (mine)
            P.X[(size_t)l * nf + lane] = xk[q];
        if (lane == 0) {
            P.chisq[l] = cur[q];
            P.best_chisq[l] = best[q];
            P.numaccept[l] = P.numaccept[l] + nacc[q];
        }
    }
}

// ---------------------------------------------------------------------------------
// The same loop with the data sum shared by NW waves on different SIMDs of the CU: with 256
// chains the chip has three of its four SIMDs idle, and a wave alone issues dependent FP64 at
// a fraction of the pipe's rate (tools/probes/mfma_f64_probe.hip), so the ~1000 cycles of the
// 16-points-per-lane chisq are the longest compute stretch of the iteration.  Wave 0 is the
// chain -- poll, proposal, Metropolis, publish, exactly as in lean_demc_kernel; waves 1..NW-1
// are helpers that sleep at a barrier until wave 0 has put the proposed coefficients in LDS,
// compute the partial sums of their virtual lanes (thread t owns virtual lanes t + 64 NW j:
// the canonical order of DataTile<NW>, so the result is bit for bit the one-wave kernel's)
// and hand them back through LDS.  Two raw barriers per iteration (LDS traffic only: no
// vector-memory drain), no flag polling: a waiting wave costs its SIMD nothing.
// NT > 1 (round 5, the mid-range of tools/mid_range.py): 2 049 ... 16 384 points.  One wave keeps
// two 1024-point tiles in registers and STREAMS the rest from L2 every iteration (3.0 us at
// 4096 points); from 4097 points on a context used to leave the hand-laid-out loops altogether
// (the general chain with four waves: 4.7 us at 8192 points, then the one-launch tiled plan at
// its flat ~6.7 us).  Here NW waves keep NT tiles between them for the whole launch -- thread
// t owns virtual lanes t + 64 NW j of EVERY tile, a virtual lane adds up its points tile after
// tile (the canonical order of mc3hip_plan.tile = 0: bit for bit what the streamed loop and
// the general chain compute) -- four waves up to eight tiles, eight waves up to sixteen (one
// workgroup per CU: 256 chains).  Tiles beyond the data are weight 0 (fma(0, 0, acc) = acc).
template <int MT, int NW, int NT = 1>
__global__ __launch_bounds__(64 * NW) void lean_demc_wide_kernel(const DevProblem P, const int nref)
{
    static_assert(NW == 2 || NW == 4 || NW == 8, "2, 4 or 8 waves share the data sum");
    constexpr int ZS = ZROW;
    constexpr int NV = 16 / NW;
    __shared__ double s_part[64 * NW];
    __shared__ double s_cf[8];
    __shared__ int s_cmd[2];                   // 0: compute, 1: nothing this time, 2: leave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = blockIdx.x;
    const int c = P.chain0 + lc;
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(
            &ctl_rw[CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
        return;
    // this thread's share of the 1024-point tiles: virtual lanes lane + 64 (wave + NW v)
    DataTile<NW, MT> tile[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        if (q * VLANES < P.npad) {
            tile[q].load(P, q * VLANES);
        } else {                                       // (no such tile: weight 0)
            tile[q].clear();
        }
    }
    auto partial = [&](const double (&cf)[MT]) {
        double acc[NV];
#pragma unroll
        for (int q = 0; q < NT; ++q) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                double m = cf[MT - 1];
#pragma unroll
                for (int j = MT - 2; j >= 0; --j)
                    m = mc3_fma(m, tile[q].x[v], cf[j]);
                const double r = (m - tile[q].d[v]) * tile[q].w[v];
                acc[v] = mc3_fma(r, r, q == 0 ? 0.0 : acc[v]);
            }
        }
#pragma unroll
        for (int s = NV / 2; s >= 1; s >>= 1) {
#pragma unroll
            for (int v = 0; v < s; ++v)
                acc[v] = acc[v] + acc[v + s];
        }
        return acc[0];
    };
    if (wave != 0) {
        // ---- helper waves -------------------------------------------------------------
        for (;;) {
            block_sync<NW>();                          // (A) the order of this iteration
            const int cmd = s_cmd[0];
            if (cmd == 2)
                return;
            if (cmd == 0) {
                double cf[MT];
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    cf[j] = s_cf[j];
                s_part[tid] = partial(cf);
            }
            block_sync<NW>();                          // (B) the partial sums are in LDS
        }
    }
    // ---- wave 0: the chain ------------------------------------------------------------
    const bool mine = lane < MT;
    const int kk = mine ? lane : 0;
    const long long R0 = P.ctl[0];
    const int n = P.nchains;
    const size_t blockstride = (size_t)n * ZS;
    const double *pbase = (R0 == 0) ? P.snap0 : P.Z + (size_t)(P.M0 + (R0 - 1) * n) * ZS;
    double *zblock = P.Z + (size_t)(P.M0 + R0 * n) * ZS;
    double *lp_row = P.log_post + (P.M0 + R0 * n + c);
    int32_t *zc_row = P.zchain + (P.M0 + R0 * n + c);
    double xk = mine ? pbase[(size_t)c * ZS + lane] : 0.0;
    double cur = P.chisq[lc];
    double best = P.best_chisq[lc];
    const long long nacc0 = P.numaccept[lc];
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    const double gamma = P.gamma;
    const long long abort_at = (P.test_abort >= 0 && lc == 0) ? (long long)P.test_abort : -1;
    int nacc = 0;
    double lpbuf = 0.0;
    Draw2 d = draw2(P.seed, (uint64_t)R0, (uint32_t)c, 0u);
    double u = u01(d.q1);
    double acc_lo, acc_hi;
    accept_window(u, acc_lo, acc_hi);
    int r1, r2;
    demc_partners(d.q0, n, c, r1, r2);
    int off1 = r1 * ZS + kk, off2 = r2 * ZS + kk;
    const long long Rend = R0 + nref;
    long long rdone = R0;
    auto flush_columns = [&](int count) {
        if (lane < count) {
            lp_row[(size_t)lane * n] = lpbuf;
            zc_row[(size_t)lane * n] = c;
        }
    };
    auto order = [&](int cmd) {                        // tell the helpers and meet them at (A)
        if (lane == 0)
            s_cmd[0] = cmd;
        block_sync<NW>();
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (long long R = R0; R < Rend; ++R) {
        if (__builtin_expect(R == abort_at, 0)) {                                         // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        const double *const addr[2] = {pbase + off1, pbase + off2};
        double val[2];
        double ring[1][2];
        ring[0][0] = load_published<0>(addr[0]);
        ring[0][1] = load_published<0>(addr[1]);
        const double u_now = u, lo_now = acc_lo, hi_now = acc_hi;
        __builtin_amdgcn_sched_barrier(0);
        d = draw2(P.seed, (uint64_t)(R + 1), (uint32_t)c, 0u);
        pin_scalar(d.q0);
        pin_scalar(d.q1);
        u = u01(d.q1);
        accept_window(u, acc_lo, acc_hi);
        demc_partners(d.q0, n, c, r1, r2);
        off1 = r1 * ZS + kk;
        off2 = r2 * ZS + kk;
        asm volatile("; pin" : "+v"(u), "+v"(off1), "+v"(off2), "+v"(acc_lo), "+v"(acc_hi));
        __builtin_amdgcn_sched_barrier(0);
        if (__builtin_expect(!poll_ring<1, 0>(addr, ring, val, ctl_rw), 0))
            break;
        const double npk = xk + (gamma * (val[0] - val[1]) + P.fepsilon * 0.0);
        const bool outk = mine && (npk < pmin_k || npk > pmax_k);
        if (__builtin_expect(!__any(outk), 1)) {
            if (mine)
                s_cf[lane] = npk;
            order(0);                                  // (A)
            // (nothing of this wave's share above the barrier: gauss.h SumHelpers::sum)
            __builtin_amdgcn_sched_barrier(0);
            double cf[MT];
#pragma unroll
            for (int j = 0; j < MT; ++j)
                cf[j] = read_lane(npk, j);
            const double t0 = partial(cf);
            block_sync<NW>();                          // (B)
            double p[NW];
            p[0] = t0;
#pragma unroll
            for (int w = 1; w < NW; ++w)
                p[w] = s_part[lane + 64 * w];
#pragma unroll
            for (int s = NW / 2; s >= 1; s >>= 1) {
#pragma unroll
                for (int w = 0; w < s; ++w)
                    p[w] = p[w] + p[w + s];
            }
            double nextchisq = wave_sum(p[0]);
            if (__builtin_expect((mc3_bits(nextchisq) & 0x7ff0000000000000ull) ==
                                 0x7ff0000000000000ull, 0)) {
                // the "+inf model => chisq = inf" trap (chain.py:322-323), resolved lazily as in
                // block_gauss_model: only a non-finite total has the model values scanned
                Model<MT> model;
                model.init_from_lanes(P, npk);
                int saw = 0;
                for (int i = lane; i < P.ndata; i += 64)
                    saw |= is_pos_inf(model.at(i));
                if (__any(saw))
                    nextchisq = mc3_from_bits(0x7ff0000000000000ull);
            }
            const bool acc = metropolis_accept(0.5 * (cur - nextchisq), u_now, lo_now, hi_now);
            xk = acc ? npk : xk;
            if (__builtin_expect(__any(acc && nextchisq < best), 0)) {
                best = nextchisq;
                if (mine)
                    P.bestx[(size_t)lc * MT + lane] = xk;
                if (lane == 0)
                    P.best_iter[lc] = R;
            }
            cur = acc ? nextchisq : cur;
            nacc += acc ? 1 : 0;
        } else {
            if (outk)
                atomicAdd(&P.outbounds[lane], 1ull);
            order(1);                                  // (A): nothing to compute
            block_sync<NW>();                          // (B)
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (mine)
            __hip_atomic_store(zblock + (size_t)c * ZS + lane, xk, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        const int slot = (int)(R - R0) & 63;
        lpbuf = (lane == slot) ? -0.5 * cur : lpbuf;
        if (__builtin_expect(slot == 63, 0)) {
            flush_columns(64);
            lp_row += (size_t)64 * n;
            zc_row += (size_t)64 * n;
        }
        pbase = zblock;
        zblock += blockstride;
        rdone = R + 1;
    }
    order(2);                                          // the helpers leave
    flush_columns((int)(rdone - R0) & 63);
    if (mine)
        P.X[(size_t)lc * MT + lane] = xk;
    if (lane == 0) {
        P.chisq[lc] = cur;
        P.best_chisq[lc] = best;
        P.numaccept[lc] = nacc0 + nacc;
    }
}

}  // namespace mc3hip
