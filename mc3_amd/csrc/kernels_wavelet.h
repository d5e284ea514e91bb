// kernels_wavelet.h -- the wavelet likelihood with a wave of its own for the chain.
//
// In persist_kernel<8, MT, true> the eight waves that share a chain's pyramid also take turns
// at everything else: wave 0 runs the proposal, stages B and C, the final sums, the
// Metropolis test and the publish while seven waves wait at barriers; the last wave computes
// the noise terms, and two more barriers hand those and the stage-A sums over.  Measured on
// C4 (2^14 points, 256 chains: one workgroup per CU; tools/stamps.py c4): of ~15 300 cycles
// per iteration only ~4 800 are stage A, the part all waves work on.
//
// Here the workgroup has NINE waves.  Waves 0..7 do nothing but the pyramid (wavelet_body CW);
// wave 8 -- the chain wave -- does nothing but run the chain (run_chain PH_CHAINWAVE: the
// general one-wave persistent chain, every sampler, priors, shared and fixed parameters,
// thinning, trace), and what it does between two likelihoods it does while the others compute:
//   * the proposal of iteration i + 1 is formed from rows and draws that were asked for
//     during iteration i;
//   * the noise terms of a proposal (software exp2 / log / reciprocals, the ordered
//     log-determinant sum) are computed by the chain wave while stage A runs, and are in LDS
//     before the first halo barrier: every thread folds its stage-A sums with the weights
//     at once, and the two barriers that used to wait for them are gone;
//   * an iteration has the halo barriers of stage A plus three (proposal there / values
//     there / chisq there) instead of those plus five, and between "chisq there" and
//     "proposal there" lies only Metropolis -> Z row -> next proposal of ONE wave.
// s_barrier counts every wave of the workgroup, so the chain wave takes part in the compute
// waves' halo barriers (doing nothing in between); it gets there first by a wide margin.
// Same arithmetic in the same order: results are bit-identical to persist_kernel's.
#pragma once
#include "run_chain.h"
#include "wavelet.h"

namespace mc3hip {

constexpr int CW_GO = 0, CW_SKIP = 1, CW_STOP = 2;
// doubles of LDS behind the compute waves' areas: the chain wave's own carve() area, the
// command word, the result
constexpr int CW_OWN = 64 + MC3HIP_MAX_PARAMS + 2 + 48 + 4;
constexpr int CW_EXTRA = CW_OWN + 4;        // (= host::CW_EXTRA_DOUBLES, ctx.h)

struct WaveletHook {
    double *s_nextp;        // the compute waves' copy of the proposal
    double *s_tab;          // noise table (wavelet_noise_terms)
    int *s_cmd;
    const double *s_result;
    int nhalo;              // halo barriers of stage A (log2 B - 1)
    int M;                  // log2 ndata
    bool skip;              // this proposal is out of bounds: nothing to compute
    bool stopped;           // the compute waves have been told to leave

    // the proposed parameter vector for the compute waves
    __device__ __forceinline__ void publish(const DevProblem &P, int, const double *own_nextp)
    {
        const int lane = threadIdx.x & 63;
        if (lane < P.nparams)
            s_nextp[lane] = own_nextp[lane];
        if (lane + 64 < P.nparams)
            s_nextp[lane + 64] = own_nextp[lane + 64];
    }

    // one likelihood; false = the workgroup leaves.  `prepare`: the chain wave's own work for
    // the next iteration, done while the others compute
    template <class F>
    __device__ __forceinline__ bool tile_phase(const DevProblem &P, bool alive_in, F &&prepare)
    {
        const int cmd = !alive_in ? CW_STOP : (skip ? CW_SKIP : CW_GO);
        if ((threadIdx.x & 63) == 0)
            s_cmd[0] = cmd;
        block_sync<9>();                                // (A) the proposal (or the order) is there
        if (cmd == CW_STOP) {
            stopped = true;
            return false;
        }
        if (cmd == CW_SKIP) {
            prepare();
            block_sync<9>();                            // (S) the order has been read: see below
            return true;
        }
        wavelet_noise_terms(s_nextp, P.nparams, M, s_tab);
        for (int k = 0; k < nhalo; ++k)
            block_sync<9>();                            // the halo barriers of stage A
        block_sync<9>();                                // (1) the values stage A left
        // (now, not before: until here the compute waves meet this wave at every halo barrier,
        // and it shares its SIMD with two of them; from here on only wave 0 is busy -- stages B
        // and C, a few thousand cycles -- and nobody waits for this wave before (3))
        prepare();
        block_sync<9>();                                // (3) the chisq is there
        return true;
    }

    __device__ __forceinline__ double collect(const DevProblem &, int) const
    {
        return skip ? 0.0 : s_result[0];
    }

    template <int MT>
    __device__ __forceinline__ double finish(const DevProblem &P, double total,
                                             const double *own_nextp) const
    {
        if (is_pos_inf(total))
            return total;
        return total + prior_term(P, own_nextp);        // as block_chisq
    }
};

// ---- the chain wave, hand-structured ------------------------------------------------------
// run_chain behind WaveletHook carries every feature of the general chain (thinning, trace,
// priors, shared parameters, the resident plans' hooks) and with them ~215 scalar registers
// spilled to vector lanes: between "chisq there" and the next "proposal there" -- the part of
// an iteration in which the eight compute waves can only wait -- it spends ~3 000 cycles
// (tools/ticks.py c4cw), most of them reloading spilled scalars.  This is the same chain for
// the commonest shape (host::cw_lean_eligible: thinning 1, no trace, no Gaussian priors, no
// shared parameters), written out so that that path is: read the chisq, two compares, select,
// store the row, add the jump that is already waiting, bounds, one LDS write, barrier.
// Everything else of an iteration -- draws, log u, the support-distribution draw, snooker rows
// and the ordinary snooker jump, DEMC's partner rows -- is asked for or computed while the
// compute waves work on the likelihood (prepare).  Same arithmetic, same draws, same order
// of floating-point operations: results are bit-identical to run_chain's.
#ifdef MC3HIP_STAMPS
#define CW_TICK(v) MC3_TICK(v)
#else
#define CW_TICK(v) do { } while (0)
#endif

template <int SMP>
__device__ __forceinline__ bool lean_chain_wave(const DevProblem &P, const int nref,
                                                double *s_nextp, double *s_tab, int *s_cmd,
                                                const double *s_result, const int nhalo,
                                                const int M)
{
    const int lane = threadIdx.x & 63;
    const int lc = (int)blockIdx.x;
    const int c = P.chain0 + lc;
    // The launch arguments this loop uses, as scalars of their own.  Left inside the wide
    // register tuples the kernel arguments arrive in they are spilled and reloaded sixteen
    // registers at a time for the sake of one (this kernel's scalar registers are full: the
    // general chain and the compute path live in it too).
    double *Zr = own_global(P.Z), *log_post = own_global(P.log_post);
    double *bestx = own_global(P.bestx);
    const double *snap0 = own_global(P.snap0);
    int32_t *zchain = own_global(P.zchain);
    long long *best_iter = own_global(P.best_iter);
    long long M0 = P.M0;
    uint64_t seed = P.seed;
    double gamma = P.gamma, fepsilon = P.fepsilon;
    int nchains = P.nchains, nparams = P.nparams, nf = P.nfree, zs = P.zs;
    asm volatile("; own" : "+s"(M0), "+s"(seed));
    asm volatile("; own" : "+s"(gamma), "+s"(fepsilon), "+s"(nchains), "+s"(nparams), "+s"(nf),
                           "+s"(zs));
    const bool mine = lane < nf;
    const int kk = mine ? lane : 0;
    const long long R0 = P.ctl[0];
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    const double *snap_first = (R0 == 0) ? snap0
                                         : Zr + (size_t)(M0 + (R0 - 1) * nchains) * zs;
    double xk = mine ? snap_first[(size_t)c * zs + lane] : 0.0;
    double cur = P.chisq[lc];
    double best = P.best_chisq[lc];
    const long long nacc0 = P.numaccept[lc];
    long long nacc = 0;
    unsigned long long oob = 0;                        // out-of-bounds proposals of this lane's parameter
    const int jfree = mine ? P.ifree[lane] : 0;
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    const bool need_normal = (SMP == MC3HIP_MRW) || (fepsilon != 0.0);
    const double pstep_k = (mine && need_normal) ? P.fstep[lane] : 0.0;
    const long long abort_at = (P.test_abort >= 0 && lc == 0) ? (long long)P.test_abort : -1;
    // the fixed parameters never change: written once, the free ones every iteration
    if (lane < nparams)
        s_nextp[lane] = P.params[lane];
    if (lane + 64 < nparams)
        s_nextp[lane + 64] = P.params[lane + 64];

    auto normal_draw = [&](uint64_t it) {              // chain.py:185, as run_chain's
        double nk = 0.0;
        if (mine) {
            const Draw2 d = draw2(seed, it, (uint32_t)c, (uint32_t)(3 + (lane >> 1)));
            const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(d.q0)));
            double sn, cs;
            mc3_det_sincos2pi(u01(d.q1), &sn, &cs);
            nk = (rad * ((lane & 1) ? sn : cs)) * pstep_k;
        }
        return nk;
    };
    // rows asked for ahead: snooker's three (chain.py:195-205), DEMC's two (chain.py:221-232)
    constexpr int NR = (SMP == MC3HIP_SNOOKER) ? 3 : 2;
    const double *ra[NR] = {};
    double rv[NR] = {};
    auto request_rows = [&](const Draws &D, long long R) {
        if constexpr (SMP == MC3HIP_SNOOKER) {
            const long long zsz = M0 + R * nchains;
            const long long iR1 = (long long)bounded(D.d0.q0, (uint64_t)zsz);
            long long iR2 = 1 + (long long)bounded(D.d0.q1, (uint64_t)(zsz - 1));
            if (iR2 == iR1)
                iR2 = 0;
            const long long iz = (long long)bounded(D.d1.q1, (uint64_t)zsz);
            ra[0] = Zr + (size_t)iR1 * zs + kk;
            ra[1] = Zr + (size_t)iR2 * zs + kk;
            ra[2] = Zr + (size_t)iz * zs + kk;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                rv[q] = load_published<false>(ra[q]);
        } else if constexpr (SMP == MC3HIP_DEMC) {
            // (the rows of refresh R - 1: asked for while their chains may still be computing
            // them -- what is not there yet comes back as the sentinel and is asked for again)
            const double *snap = (R == 0) ? snap0
                                          : Zr + (size_t)(M0 + (R - 1) * nchains) * zs;
            int r1, r2;
            demc_partners(D.d0.q0, nchains, c, r1, r2);
            ra[0] = snap + (size_t)r1 * zs + kk;
            ra[1] = snap + (size_t)r2 * zs + kk;
            rv[0] = load_published<false>(ra[0]);
            rv[1] = load_published<false>(ra[1]);
        }
    };
    DrawBatch batch;
    batch.w0 = batch.w1 = batch.w2 = batch.w3 = 0u;
    batch.base = -(long long)DRAW_BATCH - 1;           // empty
    auto draws_of = [&](long long it) {
        Draws D;
        if constexpr (SMP == MC3HIP_SNOOKER) {         // (get_draws / fill_draws, handoff.h)
            if (it < batch.base || it >= batch.base + DRAW_BATCH) {
                const uint64_t bi = (uint64_t)it + (uint64_t)((unsigned)lane / 3u);
                uint32_t w[4];
                philox4x32_10((uint32_t)bi, (uint32_t)(bi >> 32), (uint32_t)c,
                              (unsigned)lane % 3u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
                batch.w0 = w[0];
                batch.w1 = w[1];
                batch.w2 = w[2];
                batch.w3 = w[3];
                batch.base = it;
            }
            const int l = 3 * (int)(it - batch.base);
            D.d0 = batch_draw(batch, l);
            D.d1 = batch_draw(batch, l + 1);
            D.d2 = batch_draw(batch, l + 2);
        } else {
            D.d0 = draw2(seed, (uint64_t)it, (uint32_t)c, 0u);
            D.d1 = D.d0;
            D.d2 = D.d0;
        }
        return D;
    };
    Draws nxt = draws_of(R0);
    request_rows(nxt, R0);
    double normal_ahead = need_normal ? normal_draw((uint64_t)R0) : 0.0;
    double jump_ahead = 0.0;
    bool jump_ready = false;                           // snooker: the ordinary jump is formed
    double acc_lo = 0.0, acc_hi = 0.0;
    bool alive = true, stopped = false;
#ifdef MC3HIP_STAMPS
    unsigned long long ta8 = 0, ta10 = 0, ta11 = 0, ta12 = 0, ta13 = 0;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0, tk_end = 0;
    unsigned long long ta14 = 0;                       // prepare
#endif

    // (everything loaded so far retired here: a load still pending at the loop head makes the
    // compiler guard its registers with a wait INSIDE the loop -- behind the row's publish)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (long long R = R0; R < R0 + nref; ++R) {
        unsigned long long tk0 = 0, tk1 = 0, tkP = 0, tkT = 0, tkC = 0, tk2 = 0, tk3 = 0;
        (void)tk0; (void)tk1; (void)tkP; (void)tkT; (void)tkC; (void)tk2; (void)tk3;
        CW_TICK(tk0);
#ifdef MC3HIP_STAMPS
        if (tk_end != 0)
            ta13 += tk0 - tk_end;
#endif
        if (R == abort_at) {                           // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            alive = false;
        }
        const Draw2 d0 = nxt.d0, d1 = nxt.d1, d2 = nxt.d2;
        const double u = (SMP == MC3HIP_SNOOKER) ? u01(d2.q1) : u01(d0.q1);   // chain.py:259
        const double normal_k = normal_ahead;
        double jump_k, zk = 0.0;
        bool sjump = false;
        if constexpr (SMP == MC3HIP_SNOOKER) {         // chain.py:195-217
            if (jump_ready) {
                jump_k = jump_ahead;
            } else {
                sjump = u01(d1.q0) < 0.1;
                alive = wait_rows<3, false>(ra, rv, ctl_rw) && alive;
                const double z1 = rv[0], z2 = rv[1];
                zk = rv[2];
                if (sjump) {
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
                    jump_k = gamma * (z1 - z2) + fepsilon * normal_k;
                }
            }
        } else if constexpr (SMP == MC3HIP_MRW) {      // chain.py:219-220
            jump_k = normal_k;
        } else {                                       // chain.py:221-232
            alive = wait_rows<2, false>(ra, rv, ctl_rw) && alive;
            jump_k = gamma * (rv[0] - rv[1]) + fepsilon * normal_k;
        }
        CW_TICK(tk1);
        // proposal and boundaries, chain.py:235-243
        const double npk = xk + jump_k;
        const bool outk = mine && (npk < pmin_k || npk > pmax_k);
        const bool anyout = __any(outk);
        if (mine)
            s_nextp[jfree] = npk;
        const int cmd = !alive ? CW_STOP : (anyout ? CW_SKIP : CW_GO);
        if (lane == 0)
            s_cmd[0] = cmd;
        CW_TICK(tkP);
        block_sync<9>();                               // (A) the proposal (or the order) is there
        if (cmd == CW_STOP) {
            stopped = true;
            break;
        }
        // while the compute waves work: everything of the NEXT iteration that needs no chain
        // state, and the logarithm this iteration's Metropolis test is decided with
        auto prepare = [&](bool evaluated) {
            unsigned long long tp0 = 0, tp2 = 0;
            (void)tp0; (void)tp2;
            CW_TICK(tp0);
            nxt = draws_of(R + 1);
            if constexpr (SMP != MC3HIP_DEMC)
                request_rows(nxt, R + 1);
            if (evaluated) {
                accept_window(u, acc_lo, acc_hi);
                // (the bounds themselves, not log u with the rest left for after the barrier)
                asm volatile("; window" : "+v"(acc_lo), "+v"(acc_hi));
            }
            if (need_normal)
                normal_ahead = normal_draw((uint64_t)(R + 1));
            if constexpr (SMP == MC3HIP_SNOOKER) {
                // the next jump itself when it is an ordinary one (nine in ten) and its two
                // rows are there: the likelihood takes longer than their round trip
                jump_ready = false;
                if (!(u01(nxt.d1.q0) < 0.1)) {
                    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
                    const bool there = !__any(mc3_bits(rv[0]) == ROW_PENDING ||
                                              mc3_bits(rv[1]) == ROW_PENDING);
                    if (there) {
                        jump_ahead = gamma * (rv[0] - rv[1]) + fepsilon * normal_ahead;
                        jump_ready = true;
                    }
                }
            }
            CW_TICK(tp2);
#ifdef MC3HIP_STAMPS
            if (evaluated)
                ta14 += tp2 - tp0;
#endif
        };
        bool accepted = false;
        if (cmd == CW_SKIP) {
            oob += outk ? 1ull : 0ull;                 // chain.py:242-243
            prepare(false);
            block_sync<9>();                           // (S) the order has been read
            CW_TICK(tk3);
        } else {
            // (An experiment that lost: whenever the next jump is known during the likelihood
            // -- MRW always, snooker nine times in ten -- the next proposal is one of two
            // vectors, and the noise terms of both can be computed then instead of here, where
            // they take issue slots from two compute waves on this wave's SIMD: 0.2 us of a
            // 4.8 us iteration to be had.  But the terms are ~2 600 cycles of one wave's
            // dependent arithmetic, on top of ~2 000 for the rest of prepare(), and the tail
            // that hides them lasts ~3 300: C4 went from 4.8 to 5.4 us.)
            wavelet_noise_terms(s_nextp, nparams, M, s_tab);
            for (int k = 0; k < nhalo; ++k)
                block_sync<9>();                       // the halo barriers of stage A
            block_sync<9>();                           // (1) the values stage A left
            prepare(true);
            if constexpr (SMP != MC3HIP_DEMC)
                __builtin_amdgcn_s_waitcnt(0x0F70);    // (what was asked for has arrived: said
                                                       // here, the wait cannot land behind the
                                                       // stores of the publish)
            block_sync<9>();                           // (3) the chisq is there
            CW_TICK(tkT);
            const double total = s_result[0];
            CW_TICK(tkC);
            // (as WaveletHook::finish: the prior term of a problem without priors is 0.0)
            const double nextchisq = is_pos_inf(total) ? total : total + 0.0;
            CW_TICK(tk2);
            if (!sjump) {
                accepted = metropolis_accept(0.5 * (cur - nextchisq), u, acc_lo, acc_hi);
            } else {                                   // chain.py:251-255
                const double dc = xk - zk, dn = npk - zk;
                const double cnorm = lane_serial_sum(mine ? dc * dc : 0.0, nf);
                const double nnorm = lane_serial_sum(mine ? dn * dn : 0.0, nf);
                const double mrfactor = mc3_det_pow_half(nnorm / cnorm, nf - 1);
                accepted = mc3_det_exp(0.5 * (cur - nextchisq)) * mrfactor > u;
            }
            if (accepted) {                            // chain.py:257-274
                xk = npk;
                cur = nextchisq;
                nacc += 1;
            }
            CW_TICK(tk3);
        }
        // Z append, chain.py:276-289
        const long long row = M0 + R * nchains + c;
        if (mine)
            __hip_atomic_store(&Zr[(size_t)row * zs + lane], xk, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);             // publish (sc1)
        if constexpr (SMP == MC3HIP_DEMC) {
            // the partners' rows of this refresh: asked for right behind the own row, so that
            // their round trip runs while the rest of the bookkeeping does
            request_rows(nxt, R + 1);
        }
        if (lane == 0) {
            log_post[row] = -0.5 * cur;
            zchain[row] = c;
        }
        if (accepted && cur < best) {
            best = cur;
            if (mine)
                bestx[(size_t)lc * nf + lane] = xk;
            if (lane == 0)
                best_iter[lc] = R;
        }
#ifdef MC3HIP_STAMPS
        CW_TICK(tk_end);
        if (cmd == CW_GO) {
            ta8 += tk1 - tk0; tr0 += tkP - tk1; tr1 += tkT - tkP; tr2 += tkC - tkT;
            tr3 += tk2 - tkC; ta10 += tk3 - tk2; ta11 += tk_end - tk3; ta12 += 1;
        }
#endif
    }
    if (mine) {
        P.X[(size_t)lc * nf + lane] = xk;
        if (oob)
            atomicAdd(&P.outbounds[lane], oob);
    }
    if (lane == 0) {
        P.chisq[lc] = cur;
        P.best_chisq[lc] = best;
        P.numaccept[lc] = nacc0 + nacc;
    }
#ifdef MC3HIP_STAMPS
    if (lane == 0 && lc == 0) {
        P.stamps[16 + 8] += ta8; P.stamps[16 + 10] += ta10; P.stamps[16 + 11] += ta11;
        P.stamps[16 + 12] += ta12; P.stamps[16 + 13] += ta13;
        P.stamps[16 + 14] += ta14;
        P.stamps[16] += tr0; P.stamps[17] += tr1; P.stamps[18] += tr2; P.stamps[19] += tr3;
    }
#endif
    return stopped;
}

// The same chain wave for SEVERAL chains per workgroup (more chains than CUs; P.mc_chains > 0,
// wavelet_cw_kernel<.., MCK = true>).  A function of its own -- and a kernel and a translation
// unit of their own, k_wavelet_w8mc.hip -- rather than a parameter of lean_chain_wave: the
// single-chain loop above lost 17 % (C4 3.9 -> 4.6 us) when the two were one piece of code,
// although every difference folds away at compile time; this code reacts to how it is laid
// out.  What differs: the chains of a workgroup take turns inside every refresh (run_chain MC),
// their states wait in LDS between turns, and what is prepared during a likelihood is prepared
// for the next TURN -- another chain.
template <int SMP>
__device__ __forceinline__ bool lean_chain_wave_mc(const DevProblem &P, const int nref,
                                                double *s_nextp, double *s_tab, int *s_cmd,
                                                const double *s_result, const int nhalo,
                                                const int M, double *slots)
{
    const int lane = threadIdx.x & 63;
    // Several chains per workgroup (P.mc_chains > 0: more chains than CUs): chains b, b + G,
    // ... take turns inside every refresh, as in run_chain MC; what a chain carries from one
    // refresh to the next waits in LDS (`slots`: [state (zs) | chisq | best | accepts] per
    // turn), and what is prepared ahead during a likelihood is prepared for the NEXT TURN.
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    constexpr bool mc = true;
    int nturns = 1;
    if (mc) {
        nturns = 0;
        for (int q = 0; q < P.mc_chains; ++q)
            nturns += (b + q * G < P.nlocal) ? 1 : 0;
    }
    int lc = b;
    int c = P.chain0 + lc;
    // The launch arguments this loop uses, as scalars of their own.  Left inside the wide
    // register tuples the kernel arguments arrive in they are spilled and reloaded sixteen
    // registers at a time for the sake of one (this kernel's scalar registers are full: the
    // general chain and the compute path live in it too).
    double *Zr = own_global(P.Z), *log_post = own_global(P.log_post);
    double *bestx = own_global(P.bestx);
    const double *snap0 = own_global(P.snap0);
    int32_t *zchain = own_global(P.zchain);
    long long *best_iter = own_global(P.best_iter);
    long long M0 = P.M0;
    uint64_t seed = P.seed;
    double gamma = P.gamma, fepsilon = P.fepsilon;
    int nchains = P.nchains, nparams = P.nparams, nf = P.nfree, zs = P.zs;
    asm volatile("; own" : "+s"(M0), "+s"(seed));
    asm volatile("; own" : "+s"(gamma), "+s"(fepsilon), "+s"(nchains), "+s"(nparams), "+s"(nf),
                           "+s"(zs));
    const bool mine = lane < nf;
    const int kk = mine ? lane : 0;
    const long long R0 = P.ctl[0];
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    const double *snap_first = (R0 == 0) ? snap0
                                         : Zr + (size_t)(M0 + (R0 - 1) * nchains) * zs;
    double xk = mine ? snap_first[(size_t)c * zs + lane] : 0.0;
    double cur = P.chisq[lc];
    double best = P.best_chisq[lc];
    long long nacc = P.numaccept[lc];
    const int sstride = zs + 4;
    if (mc) {
        for (int q = 0; q < nturns; ++q) {
            const int l = b + q * G;
            double *slot = slots + q * sstride;
            if (mine)
                slot[lane] = snap_first[(size_t)(P.chain0 + l) * zs + lane];
            if (lane == 0) {
                slot[zs] = P.chisq[l];
                slot[zs + 1] = P.best_chisq[l];
                slot[zs + 2] = mc3_from_bits((unsigned long long)P.numaccept[l]);
            }
        }
    }
    unsigned long long oob = 0;                        // out-of-bounds proposals of this lane's parameter
    const int jfree = mine ? P.ifree[lane] : 0;
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    const bool need_normal = (SMP == MC3HIP_MRW) || (fepsilon != 0.0);
    const double pstep_k = (mine && need_normal) ? P.fstep[lane] : 0.0;
    const long long abort_at = (P.test_abort >= 0 && b == 0) ? (long long)P.test_abort : -1;
    // the fixed parameters never change: written once, the free ones every iteration
    if (lane < nparams)
        s_nextp[lane] = P.params[lane];
    if (lane + 64 < nparams)
        s_nextp[lane + 64] = P.params[lane + 64];

    auto normal_draw = [&](uint64_t it, int ch) {      // chain.py:185, as run_chain's
        double nk = 0.0;
        if (mine) {
            const Draw2 d = draw2(seed, it, (uint32_t)ch, (uint32_t)(3 + (lane >> 1)));
            const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(d.q0)));
            double sn, cs;
            mc3_det_sincos2pi(u01(d.q1), &sn, &cs);
            nk = (rad * ((lane & 1) ? sn : cs)) * pstep_k;
        }
        return nk;
    };
    // rows asked for ahead: snooker's three (chain.py:195-205), DEMC's two (chain.py:221-232)
    constexpr int NR = (SMP == MC3HIP_SNOOKER) ? 3 : 2;
    const double *ra[NR] = {};
    double rv[NR] = {};
    auto request_rows = [&](const Draws &D, long long R, int ch) {
        if constexpr (SMP == MC3HIP_SNOOKER) {
            const long long zsz = M0 + R * nchains;
            const long long iR1 = (long long)bounded(D.d0.q0, (uint64_t)zsz);
            long long iR2 = 1 + (long long)bounded(D.d0.q1, (uint64_t)(zsz - 1));
            if (iR2 == iR1)
                iR2 = 0;
            const long long iz = (long long)bounded(D.d1.q1, (uint64_t)zsz);
            ra[0] = Zr + (size_t)iR1 * zs + kk;
            ra[1] = Zr + (size_t)iR2 * zs + kk;
            ra[2] = Zr + (size_t)iz * zs + kk;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                rv[q] = load_published<false>(ra[q]);
        } else if constexpr (SMP == MC3HIP_DEMC) {
            // (the rows of refresh R - 1: asked for while their chains may still be computing
            // them -- what is not there yet comes back as the sentinel and is asked for again)
            const double *snap = (R == 0) ? snap0
                                          : Zr + (size_t)(M0 + (R - 1) * nchains) * zs;
            int r1, r2;
            demc_partners(D.d0.q0, nchains, ch, r1, r2);
            ra[0] = snap + (size_t)r1 * zs + kk;
            ra[1] = snap + (size_t)r2 * zs + kk;
            rv[0] = load_published<false>(ra[0]);
            rv[1] = load_published<false>(ra[1]);
        }
    };
    DrawBatch batch;
    batch.w0 = batch.w1 = batch.w2 = batch.w3 = 0u;
    batch.base = -(long long)DRAW_BATCH - 1;           // empty
    auto draws_of = [&](long long it, int ch) {
        Draws D;
        if (SMP == MC3HIP_SNOOKER && mc) {             // (the batch below is one chain's)
            D.d0 = draw2(seed, (uint64_t)it, (uint32_t)ch, 0u);
            D.d1 = draw2(seed, (uint64_t)it, (uint32_t)ch, 1u);
            D.d2 = draw2(seed, (uint64_t)it, (uint32_t)ch, 2u);
            return D;
        }
        if constexpr (SMP == MC3HIP_SNOOKER) {         // (get_draws / fill_draws, handoff.h)
            if (it < batch.base || it >= batch.base + DRAW_BATCH) {
                const uint64_t bi = (uint64_t)it + (uint64_t)((unsigned)lane / 3u);
                uint32_t w[4];
                philox4x32_10((uint32_t)bi, (uint32_t)(bi >> 32), (uint32_t)c,
                              (unsigned)lane % 3u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
                batch.w0 = w[0];
                batch.w1 = w[1];
                batch.w2 = w[2];
                batch.w3 = w[3];
                batch.base = it;
            }
            const int l = 3 * (int)(it - batch.base);
            D.d0 = batch_draw(batch, l);
            D.d1 = batch_draw(batch, l + 1);
            D.d2 = batch_draw(batch, l + 2);
        } else {
            D.d0 = draw2(seed, (uint64_t)it, (uint32_t)ch, 0u);
            D.d1 = D.d0;
            D.d2 = D.d0;
        }
        return D;
    };
    Draws nxt = draws_of(R0, c);
    request_rows(nxt, R0, c);
    double normal_ahead = need_normal ? normal_draw((uint64_t)R0, c) : 0.0;
    double jump_ahead = 0.0;
    bool jump_ready = false;                           // snooker: the ordinary jump is formed
    double acc_lo = 0.0, acc_hi = 0.0;
    bool alive = true, stopped = false;
#ifdef MC3HIP_STAMPS
    unsigned long long ta8 = 0, ta10 = 0, ta11 = 0, ta12 = 0, ta13 = 0;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0, tk_end = 0;
    unsigned long long ta14 = 0;                       // prepare
#endif

    for (long long R = R0; R < R0 + nref && !stopped; ++R) {
    for (int q = 0; q < nturns; ++q) {
        if (mc) {                                      // this turn's chain
            lc = b + q * G;
            c = P.chain0 + lc;
            const double *slot = slots + q * sstride;
            xk = mine ? slot[lane] : 0.0;
            cur = slot[zs];
            best = slot[zs + 1];
            nacc = (long long)mc3_bits(slot[zs + 2]);
        }
        // the turn after this one: the next chain in this refresh, or the first in the next
        const bool last_turn = q + 1 == nturns;
        const long long R2 = last_turn ? R + 1 : R;
        const int c2 = mc ? P.chain0 + b + (last_turn ? 0 : q + 1) * G : c;
        unsigned long long tk0 = 0, tk1 = 0, tkP = 0, tkT = 0, tkC = 0, tk2 = 0, tk3 = 0;
        (void)tk0; (void)tk1; (void)tkP; (void)tkT; (void)tkC; (void)tk2; (void)tk3;
        CW_TICK(tk0);
#ifdef MC3HIP_STAMPS
        if (tk_end != 0)
            ta13 += tk0 - tk_end;
#endif
        if (R == abort_at && q == 0) {                 // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            alive = false;
        }
        const Draw2 d0 = nxt.d0, d1 = nxt.d1, d2 = nxt.d2;
        const double u = (SMP == MC3HIP_SNOOKER) ? u01(d2.q1) : u01(d0.q1);   // chain.py:259
        const double normal_k = normal_ahead;
        double jump_k, zk = 0.0;
        bool sjump = false;
        if constexpr (SMP == MC3HIP_SNOOKER) {         // chain.py:195-217
            if (jump_ready) {
                jump_k = jump_ahead;
            } else {
                sjump = u01(d1.q0) < 0.1;
                alive = wait_rows<3, false>(ra, rv, ctl_rw) && alive;
                const double z1 = rv[0], z2 = rv[1];
                zk = rv[2];
                if (sjump) {
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
                    jump_k = gamma * (z1 - z2) + fepsilon * normal_k;
                }
            }
        } else if constexpr (SMP == MC3HIP_MRW) {      // chain.py:219-220
            jump_k = normal_k;
        } else {                                       // chain.py:221-232
            alive = wait_rows<2, false>(ra, rv, ctl_rw) && alive;
            jump_k = gamma * (rv[0] - rv[1]) + fepsilon * normal_k;
        }
        CW_TICK(tk1);
        // proposal and boundaries, chain.py:235-243
        const double npk = xk + jump_k;
        const bool outk = mine && (npk < pmin_k || npk > pmax_k);
        const bool anyout = __any(outk);
        if (mine)
            s_nextp[jfree] = npk;
        const int cmd = !alive ? CW_STOP : (anyout ? CW_SKIP : CW_GO);
        if (lane == 0)
            s_cmd[0] = cmd;
        CW_TICK(tkP);
        block_sync<9>();                               // (A) the proposal (or the order) is there
        if (cmd == CW_STOP) {
            stopped = true;
            break;
        }
        // while the compute waves work: everything of the NEXT iteration that needs no chain
        // state, and the logarithm this iteration's Metropolis test is decided with
        auto prepare = [&](bool evaluated) {
            unsigned long long tp0 = 0, tp2 = 0;
            (void)tp0; (void)tp2;
            CW_TICK(tp0);
            nxt = draws_of(R2, c2);
            if constexpr (SMP != MC3HIP_DEMC)
                request_rows(nxt, R2, c2);
            if (evaluated) {
                accept_window(u, acc_lo, acc_hi);
                // (the bounds themselves, not log u with the rest left for after the barrier)
                asm volatile("; window" : "+v"(acc_lo), "+v"(acc_hi));
            }
            if (need_normal)
                normal_ahead = normal_draw((uint64_t)R2, c2);
            if constexpr (SMP == MC3HIP_SNOOKER) {
                // the next jump itself when it is an ordinary one (nine in ten) and its two
                // rows are there: the likelihood takes longer than their round trip
                jump_ready = false;
                if (!(u01(nxt.d1.q0) < 0.1)) {
                    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
                    const bool there = !__any(mc3_bits(rv[0]) == ROW_PENDING ||
                                              mc3_bits(rv[1]) == ROW_PENDING);
                    if (there) {
                        jump_ahead = gamma * (rv[0] - rv[1]) + fepsilon * normal_ahead;
                        jump_ready = true;
                    }
                }
            }
            CW_TICK(tp2);
#ifdef MC3HIP_STAMPS
            if (evaluated)
                ta14 += tp2 - tp0;
#endif
        };
        bool accepted = false;
        if (cmd == CW_SKIP) {
            oob += outk ? 1ull : 0ull;                 // chain.py:242-243
            prepare(false);
            block_sync<9>();                           // (S) the order has been read
            CW_TICK(tk3);
        } else {
            // (An experiment that lost: whenever the next jump is known during the likelihood
            // -- MRW always, snooker nine times in ten -- the next proposal is one of two
            // vectors, and the noise terms of both can be computed then instead of here, where
            // they take issue slots from two compute waves on this wave's SIMD: 0.2 us of a
            // 4.8 us iteration to be had.  But the terms are ~2 600 cycles of one wave's
            // dependent arithmetic, on top of ~2 000 for the rest of prepare(), and the tail
            // that hides them lasts ~3 300: C4 went from 4.8 to 5.4 us.)
            wavelet_noise_terms(s_nextp, nparams, M, s_tab);
            for (int k = 0; k < nhalo; ++k)
                block_sync<9>();                       // the halo barriers of stage A
            block_sync<9>();                           // (1) the values stage A left
            prepare(true);
            if constexpr (SMP != MC3HIP_DEMC)
                __builtin_amdgcn_s_waitcnt(0x0F70);    // (what was asked for has arrived: said
                                                       // here, the wait cannot land behind the
                                                       // stores of the publish)
            block_sync<9>();                           // (3) the chisq is there
            CW_TICK(tkT);
            const double total = s_result[0];
            CW_TICK(tkC);
            // (as WaveletHook::finish: the prior term of a problem without priors is 0.0)
            const double nextchisq = is_pos_inf(total) ? total : total + 0.0;
            CW_TICK(tk2);
            if (!sjump) {
                accepted = metropolis_accept(0.5 * (cur - nextchisq), u, acc_lo, acc_hi);
            } else {                                   // chain.py:251-255
                const double dc = xk - zk, dn = npk - zk;
                const double cnorm = lane_serial_sum(mine ? dc * dc : 0.0, nf);
                const double nnorm = lane_serial_sum(mine ? dn * dn : 0.0, nf);
                const double mrfactor = mc3_det_pow_half(nnorm / cnorm, nf - 1);
                accepted = mc3_det_exp(0.5 * (cur - nextchisq)) * mrfactor > u;
            }
            if (accepted) {                            // chain.py:257-274
                xk = npk;
                cur = nextchisq;
                nacc += 1;
            }
            CW_TICK(tk3);
        }
        // Z append, chain.py:276-289
        const long long row = M0 + R * nchains + c;
        if (mine)
            __hip_atomic_store(&Zr[(size_t)row * zs + lane], xk, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);             // publish (sc1)
        if constexpr (SMP == MC3HIP_DEMC) {
            // the next turn's partner rows: asked for right behind the own row, so that their
            // round trip runs while the rest of the bookkeeping does
            request_rows(nxt, R2, c2);
        }
        if (lane == 0) {
            log_post[row] = -0.5 * cur;
            zchain[row] = c;
        }
        if (accepted && cur < best) {
            best = cur;
            if (mine)
                bestx[(size_t)lc * nf + lane] = xk;
            if (lane == 0)
                best_iter[lc] = R;
        }
        if (mc) {                                      // this chain waits for its next turn
            double *slot = slots + q * sstride;
            if (mine)
                slot[lane] = xk;
            if (lane == 0) {
                slot[zs] = cur;
                slot[zs + 1] = best;
                slot[zs + 2] = mc3_from_bits((unsigned long long)nacc);
            }
        }
#ifdef MC3HIP_STAMPS
        CW_TICK(tk_end);
        if (cmd == CW_GO) {
            ta8 += tk1 - tk0; tr0 += tkP - tk1; tr1 += tkT - tkP; tr2 += tkC - tkT;
            tr3 += tk2 - tkC; ta10 += tk3 - tk2; ta11 += tk_end - tk3; ta12 += 1;
        }
#endif
    }   // turns
    }   // refreshes
    if (mine && oob)
        atomicAdd(&P.outbounds[lane], oob);
    if (mc) {
        for (int q = 0; q < nturns; ++q) {
            const int l = b + q * G;
            const double *slot = slots + q * sstride;
            if (mine)
                P.X[(size_t)l * nf + lane] = slot[lane];
            if (lane == 0) {
                P.chisq[l] = slot[zs];
                P.best_chisq[l] = slot[zs + 1];
                P.numaccept[l] = (long long)mc3_bits(slot[zs + 2]);
            }
        }
    } else {
        if (mine)
            P.X[(size_t)lc * nf + lane] = xk;
        if (lane == 0) {
            P.chisq[lc] = cur;
            P.best_chisq[lc] = best;
            P.numaccept[lc] = nacc;
        }
    }
#ifdef MC3HIP_STAMPS
    if (lane == 0 && b == 0) {
        P.stamps[16 + 8] += ta8; P.stamps[16 + 10] += ta10; P.stamps[16 + 11] += ta11;
        P.stamps[16 + 12] += ta12; P.stamps[16 + 13] += ta13;
        P.stamps[16 + 14] += ta14;
        P.stamps[16] += tr0; P.stamps[17] += tr1; P.stamps[18] += tr2; P.stamps[19] += tr3;
    }
#endif
    return stopped;
}

// (SMP: the sampler, known at compile time -- the chain wave's path between two likelihoods is
// what an iteration waits for, so the other samplers' branches are compiled out of it)
// (MCK: the instance for several chains per workgroup, lean_chain_wave_mc.  GEN: the instance
// whose ninth wave runs the general chain -- trace, priors, shared parameters, thinning -- so
// that the kernel of the common shape carries only the hand-structured one)
template <int MT, int SMP, bool MCK = false, bool GEN = false>
__global__ __launch_bounds__(576) void wavelet_cw_kernel(const DevProblem P, const int nref)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SmemLayout L = carve(smem, P, 512);           // the compute waves' areas
    double *own = L.xs + (P.ndata > 128 ? P.ndata : 128);
    int *s_cmd = reinterpret_cast<int *>(own + CW_OWN);
    double *s_result = own + CW_OWN + 2;
    const int tid = threadIdx.x;
    // a launch queued behind one that gave up must not run: its inputs are missing
    if (launch_given_up(P.ctl, s_cmd + 3))             // (one decision per workgroup)
        return;
    if (tid < 512)
        wavelet_stage<8, MT>(P, L.xs);
    block_sync<9>();
    const int B = P.ndata / 512;                        // 16 or 32 (mc3hip.hip choose_wpc)
    if (tid >= 512) {
        // ---- the chain wave ---------------------------------------------------------------
        WaveletHook H;
        H.s_nextp = L.nextp;
        H.s_tab = L.tab;
        H.s_cmd = s_cmd;
        H.s_result = s_result;
        H.nhalo = ilog2(B) - 1;
        H.M = ilog2(P.ndata);
        H.skip = false;
        H.stopped = false;
        if constexpr (MCK) {
            H.stopped = lean_chain_wave_mc<SMP>(P, nref, L.nextp, L.tab, s_cmd, s_result, H.nhalo,
                                                H.M, smem + P.mc_off);
        } else if constexpr (GEN) {
            run_chain<1, MT, false, false, true, SMP, false, PH_CHAINWAVE, false, 0, false,
                      WaveletHook>(P, P.ctl[0], nref, 0, -1, &H, (int)blockIdx.x, own);
        } else {
            H.stopped = lean_chain_wave<SMP>(P, nref, L.nextp, L.tab, s_cmd, s_result, H.nhalo,
                                             H.M);
        }
        if (!H.stopped) {                               // the run is over: the others leave too
            if ((tid & 63) == 0)
                s_cmd[0] = CW_STOP;
            block_sync<9>();
        }
        return;
    }
    // ---- the compute waves -------------------------------------------------------------------
    // (the ninth wave shares a SIMD with two of these and computes the proposal's noise terms
    // while they run level 0; raising these waves' priority above its own with
    // s_setprio measured: nothing)
    for (;;) {
        block_sync<9>();                                // (A)
        MC3_LSTAMP(0);
        // (the order and the coefficients in one trip to LDS, not one after the other)
        const int cmd = s_cmd[0];
        Model<MT> model;
        model.init(P, L.nextp);
        if (cmd == CW_STOP)
            return;
        if (cmd == CW_SKIP) {
            // (S): the chain wave writes the next order only behind this barrier.  Without it a
            // skipped iteration had no barrier between this wave's read of the order and the
            // chain wave's next write -- safe only because preparing an iteration takes
            // thousands of cycles; the waves sleep here instead of at (A), at no cost
            block_sync<9>();
            continue;
        }
        double c;
        if (B == 32)
            c = wavelet_body<8, MT, 32, false, true>(P, L.nextp, L.part, L.flag, L.tab, L.halo,
                                                     L.xs, nullptr, 0, nullptr, 0, &model);
        else
            c = wavelet_body<8, MT, 16, false, true>(P, L.nextp, L.part, L.flag, L.tab, L.halo,
                                                     L.xs, nullptr, 0, nullptr, 0, &model);
        if (tid == 64)                                  // (wavelet_body CW: wave 1 runs the tail)
            s_result[0] = c;
        MC3_LSTAMP(1);
        block_sync<9>();                                // (3)
        MC3_LSTAMP(2);
        MC3_LSTAMP_FLUSH();
    }
}

// ---- several chains per workgroup, PIPELINED (snooker and mrw) -----------------------------
// wavelet_cw_kernel<.., MCK> above takes the chains of a workgroup one after the other, and of a
// turn's ~12 000 cycles (tools/pipe_diag.py) the eight compute waves work about half: they wait
// through the single-wave tail (stages B and C, the sums, the two divisions) and through the chain
// wave's turn.  With two or more chains per workgroup nothing forces that: the NEXT turn is
// another chain, whose proposal needs nothing of this likelihood.  Here the workgroup has ELEVEN
// waves and time is cut into slots of one stage A each:
//   waves 0..7   stage A of the turn issued into slot s, nothing else (wavelet_body PIPE);
//   wave 9       the tail of the turn issued into slot s - 1 (PipeTail), at once: its chisq is in
//                LDS ~2 300 cycles into the slot, and a word beside it says so;
//   wave 10      the noise terms of slot s's proposal (needed at the end of stage A);
//   wave 8       the chain: draws and row requests of the turn after the one in flight and the
//                logarithm of the Metropolis test while the tail wave works; then the test, the
//                Z row and the books of slot s - 1's turn; then the proposal for slot s + 1.
// All of that happens during level 0 of stage A, the long stretch (two compute waves per SIMD:
// ~3 700 cycles alone, ~5 000 beside the helpers, which sit on three different SIMDs); the helpers
// then only attend the halo barriers (s_barrier counts every wave).  Proposals, noise tables and
// orders alternate between two LDS buffers (slot parity).  A turn can only be issued when its
// chain's previous turn has been decided (always so with two chains or more) and its history rows
// are there; a row that is NOT there may be the one this workgroup still owes (the turn in
// flight), so the chain wave never waits with a turn in flight: it issues an empty slot (a bubble
// -- the tail and the decision of the turn in flight run, the compute waves idle) and asks
// again.  A workgroup that blocks has therefore published everything it can: no cycle.
// Same arithmetic, same draws, same order per chain: results are bit-identical.
// Measured (MI355X, 2^14 points, snooker): 512 chains 8.12 -> 6.0 us per iteration, 1024 chains
// 15.4 -> 11.9; every proposal evaluated (c4x2wide) 10.0 -> 7.0; mrw 9.1 -> 7.2.
// (tools/pipe_diag.py: a diagnostic build in which the chain wave and the tail wave of workgroup 0
// add up, per segment of a slot, the cycles they work and the cycles the segment lasts)
#ifdef MC3HIP_PIPE_DIAG
#define PIPE_SYNC(i)                                                                     \
    do {                                                                                  \
        const unsigned long long a_ = (10ull * __builtin_amdgcn_s_memrealtime());                       \
        dg_work[i] += a_ - dg_t0;                                                         \
        block_sync<9>();                                                                  \
        const unsigned long long b_ = (10ull * __builtin_amdgcn_s_memrealtime());                       \
        dg_dur[i] += b_ - dg_t0;                                                          \
        dg_t0 = b_;                                                                       \
    } while (0)
#define PIPE_DIAG_DECL                                                                    \
    unsigned long long dg_work[6] = {0, 0, 0, 0, 0, 0}, dg_dur[6] = {0, 0, 0, 0, 0, 0};   \
    unsigned long long dg_part[5] = {0, 0, 0, 0, 0}, dg_p0 = 0;                           \
    unsigned long long dg_t0 = (10ull * __builtin_amdgcn_s_memrealtime()), dg_n = 0
#define PIPE_PART(i)                                                                      \
    do {                                                                                  \
        const unsigned long long a_ = (10ull * __builtin_amdgcn_s_memrealtime());                       \
        if ((i) >= 0)                                                                     \
            dg_part[(i) < 0 ? 0 : (i)] += a_ - dg_p0;                                     \
        dg_p0 = a_;                                                                       \
    } while (0)
#define PIPE_DIAG_PRINT(who)                                                              \
    do {                                                                                  \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)                                   \
            printf("%s slots %llu | work %llu %llu %llu %llu %llu %llu | lasts %llu %llu %llu %llu %llu %llu\n", \
                   who, dg_n, dg_work[0] / dg_n, dg_work[1] / dg_n, dg_work[2] / dg_n,    \
                   dg_work[3] / dg_n, dg_work[4] / dg_n, dg_work[5] / dg_n,               \
                   dg_dur[0] / dg_n, dg_dur[1] / dg_n, dg_dur[2] / dg_n, dg_dur[3] / dg_n, \
                   dg_dur[4] / dg_n, dg_dur[5] / dg_n);                                   \
    } while (0)
#else
#define PIPE_SYNC(i) block_sync<9>()
#define PIPE_PART(i) do { } while (0)
#define PIPE_DIAG_DECL do { } while (0)
#define PIPE_DIAG_PRINT(who) do { } while (0)
#endif

template <int SMP>
__device__ __forceinline__ void pipe_chain_wave(const DevProblem &P, const int nref, double *np0,
                                                double *np1, double *tab0, double *tab1,
                                                int *s_cmd, const double *s_result,
                                                const int nhalo, const int M, double *slots)
{
    static_assert(SMP != MC3HIP_DEMC, "DEMC's partner rows are the refresh before: no slack");
    const int lane = threadIdx.x & 63;
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    int nturns = 0;
    for (int q = 0; q < P.mc_chains; ++q)
        nturns += (b + q * G < P.nlocal) ? 1 : 0;
    double *Zr = own_global(P.Z), *log_post = own_global(P.log_post);
    double *bestx = own_global(P.bestx);
    const double *snap0 = own_global(P.snap0);
    int32_t *zchain = own_global(P.zchain);
    long long *best_iter = own_global(P.best_iter);
    long long M0 = P.M0;
    uint64_t seed = P.seed;
    double gamma = P.gamma, fepsilon = P.fepsilon;
    int nchains = P.nchains, nparams = P.nparams, nf = P.nfree, zs = P.zs;
    asm volatile("; own" : "+s"(M0), "+s"(seed));
    asm volatile("; own" : "+s"(gamma), "+s"(fepsilon), "+s"(nchains), "+s"(nparams), "+s"(nf),
                           "+s"(zs));
    const bool mine = lane < nf;
    const int kk = mine ? lane : 0;
    const long long R0 = P.ctl[0];
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    const double *snap_first = (R0 == 0) ? snap0
                                         : Zr + (size_t)(M0 + (R0 - 1) * nchains) * zs;
    const int sstride = zs + 4;
    for (int q = 0; q < nturns; ++q) {
        const int l = b + q * G;
        double *slot = slots + q * sstride;
        if (mine)
            slot[lane] = snap_first[(size_t)(P.chain0 + l) * zs + lane];
        if (lane == 0) {
            slot[zs] = P.chisq[l];
            slot[zs + 1] = P.best_chisq[l];
            slot[zs + 2] = mc3_from_bits((unsigned long long)P.numaccept[l]);
        }
    }
    unsigned long long oob = 0;
    const int jfree = mine ? P.ifree[lane] : 0;
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    const bool need_normal = (SMP == MC3HIP_MRW) || (fepsilon != 0.0);
    const double pstep_k = (mine && need_normal) ? P.fstep[lane] : 0.0;
    const long long abort_at = (P.test_abort >= 0 && b == 0) ? (long long)P.test_abort : -1;
    // the fixed parameters never change: written once (both buffers), the free ones every turn
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double *np = h ? np1 : np0;
        if (lane < nparams)
            np[lane] = P.params[lane];
        if (lane + 64 < nparams)
            np[lane + 64] = P.params[lane + 64];
    }
    auto normal_draw = [&](uint64_t it, int ch) {      // chain.py:185, as run_chain's
        double nk = 0.0;
        if (mine) {
            const Draw2 d = draw2(seed, it, (uint32_t)ch, (uint32_t)(3 + (lane >> 1)));
            const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(d.q0)));
            double sn, cs;
            mc3_det_sincos2pi(u01(d.q1), &sn, &cs);
            nk = (rad * ((lane & 1) ? sn : cs)) * pstep_k;
        }
        return nk;
    };
    double rv[3] = {};
    // (the rows' addresses are a function of the draws: formed again where a row is asked for
    // again, not carried from slot to slot -- this wave has no registers to spare)
    auto row_addresses = [&](const Draws &D, long long R, const double *(&ra)[3]) {   // chain.py:195-205
        if constexpr (SMP == MC3HIP_SNOOKER) {
            const long long zsz = M0 + R * nchains;
            const long long iR1 = (long long)bounded(D.d0.q0, (uint64_t)zsz);
            long long iR2 = 1 + (long long)bounded(D.d0.q1, (uint64_t)(zsz - 1));
            if (iR2 == iR1)
                iR2 = 0;
            const long long iz = (long long)bounded(D.d1.q1, (uint64_t)zsz);
            ra[0] = Zr + (size_t)iR1 * zs + kk;
            ra[1] = Zr + (size_t)iR2 * zs + kk;
            ra[2] = Zr + (size_t)iz * zs + kk;
        }
    };
    auto request_rows = [&](const Draws &D, long long R) {
        if constexpr (SMP == MC3HIP_SNOOKER) {
            const double *ra[3];
            row_addresses(D, R, ra);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                rv[q] = load_published<false>(ra[q]);
        }
    };
    // The draws of the next 21 turns in ONE Philox call, a (turn, slot) per lane (the idea of
    // fill_draws, handoff.h, with the turn's own chain and refresh in every lane): three calls per
    // turn were ~2 000 cycles of a slot.  `n`: the turn's index since this launch began.
    DrawBatch batch;
    batch.w0 = batch.w1 = batch.w2 = batch.w3 = 0u;
    batch.base = -(long long)DRAW_BATCH - 1;           // empty
    auto draws_of = [&](long long n) {
        if (n < batch.base || n >= batch.base + DRAW_BATCH) {
            const uint32_t nl = (uint32_t)n + (uint32_t)lane / 3u;
            const uint32_t rl = nl / (uint32_t)nturns, ql = nl - rl * (uint32_t)nturns;
            const uint64_t it = (uint64_t)R0 + rl;
            uint32_t w[4];
            philox4x32_10((uint32_t)it, (uint32_t)(it >> 32),
                          (uint32_t)(P.chain0 + b + (int)ql * G), (unsigned)lane % 3u,
                          (uint32_t)seed, (uint32_t)(seed >> 32), w);
            batch.w0 = w[0];
            batch.w1 = w[1];
            batch.w2 = w[2];
            batch.w3 = w[3];
            batch.base = n;
        }
        const int l = 3 * (int)(n - batch.base);
        Draws D;
        D.d0 = batch_draw(batch, l);
        D.d1 = batch_draw(batch, l + 1);
        D.d2 = batch_draw(batch, l + 2);
        return D;
    };
    // a turn on its way through the pipeline
    struct Turn {
        bool valid, go, sjump, outk, spec;
        int q, c;
        long long R, n;
        double npk, zk, u;
    };
    Turn cur, pend, cand;                               // slot s, slot s - 1, slot s + 1
    cur.valid = pend.valid = cand.valid = false;
    // the turn to issue next, and what has been prepared for it
    long long Rn = R0, tn = 0;
    int qn = 0, cn = P.chain0 + b;
    bool more = nref > 0;
    // (its draws are read from the batch wherever they are needed: a dozen lane reads against a
    // dozen registers held from slot to slot)
    request_rows(draws_of(tn), Rn);
    double normal_ahead = need_normal ? normal_draw((uint64_t)Rn, cn) : 0.0;
    double jump_ahead = 0.0;
    bool jump_ready = false;
    bool alive = true;

    // What was asked for has arrived -- said HERE, before the decision's stores: loads and stores
    // retire through one counter, and a wait behind the Z row's store waits for that store
    // (~1 000 cycles).  The ordinary snooker jump (nine in ten) at once.
    bool rows_there = false;
    auto jump_early = [&]() {
        if constexpr (SMP == MC3HIP_SNOOKER) {
            jump_ready = false;
            rows_there = false;
            // (on every path, and the values named: the compiler then knows them settled, and puts
            // no wait of its own in front of their next use -- which lies behind those stores)
            __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0)
            asm volatile("; rows" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]));
            if (more) {
                const bool there2 = !__any(mc3_bits(rv[0]) == ROW_PENDING ||
                                           mc3_bits(rv[1]) == ROW_PENDING);
                rows_there = there2 && !__any(mc3_bits(rv[2]) == ROW_PENDING);
                if (there2 && !(u01(draws_of(tn).d1.q0) < 0.1)) {
                    jump_ahead = gamma * (rv[0] - rv[1]) + fepsilon * normal_ahead;
                    jump_ready = true;
                }
            }
        }
    };
    // the proposal of turn (Rn, qn) into `np`; the order for the slot it opens
    auto try_issue = [&](double *np) -> int {
        cand.valid = false;
        if (!alive)
            return CW_STOP;
        if (!more)
            return cur.valid ? CW_SKIP : CW_STOP;       // (the last turn's tail and decision)
        // The chain's previous turn is the one in flight (one chain in this workgroup): SPECULATE
        // ON REJECT.  Most proposals are rejected, and after a reject the next proposal starts
        // from the state this turn started from -- so it is issued as if; if the turn in flight is
        // accepted after all, this one is squashed (nothing of it has left the workgroup: its
        // likelihood runs to its end unread) and issued again from the new state.  A proposal out
        // of bounds is a reject already.
        const bool spec = cur.valid && cur.q == qn && cur.go;
        if (Rn == abort_at && qn == 0) {                // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            alive = false;
            return CW_STOP;
        }
        const double *slot = slots + qn * sstride;
        const double xk = mine ? slot[lane] : 0.0;
        const Draws nxt = draws_of(tn);
        double jump_k, zk = 0.0;
        bool sjump = false;
        if constexpr (SMP == MC3HIP_SNOOKER) {         // chain.py:195-217
            if (jump_ready) {
                jump_k = jump_ahead;
            } else {
                sjump = u01(nxt.d1.q0) < 0.1;
                if (!rows_there) {
                    if (cur.valid)
                        return CW_SKIP;                 // (never wait with a turn in flight)
                    const double *ra[3];
                    row_addresses(nxt, Rn, ra);
                    alive = wait_rows<3, false>(ra, rv, ctl_rw) && alive;
                    if (!alive)
                        return CW_STOP;
                }
                const double z1 = rv[0], z2 = rv[1];
                zk = rv[2];
                if (sjump) {
                    const double gs = 1.2 + (2.2 - 1.2) * u01(nxt.d2.q0);
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
                    jump_k = gamma * (z1 - z2) + fepsilon * normal_ahead;
                }
            }
        } else {                                        // chain.py:219-220
            jump_k = normal_ahead;
        }
        const double npk = xk + jump_k;                 // chain.py:235-243
        const bool outk = mine && (npk < pmin_k || npk > pmax_k);
        const bool anyout = __any(outk);
        if (mine)
            np[jfree] = npk;
        cand.valid = true;
        cand.spec = spec;
        cand.n = tn;
        cand.go = !anyout;
        cand.sjump = sjump;
        cand.outk = outk;
        cand.q = qn;
        cand.c = cn;
        cand.R = Rn;
        cand.npk = npk;
        cand.zk = zk;
        cand.u = (SMP == MC3HIP_SNOOKER) ? u01(nxt.d2.q1) : u01(nxt.d0.q1);   // chain.py:259
        return anyout ? CW_SKIP : CW_GO;
    };
    // Metropolis, Z append and the books of a turn whose chisq is in s_result
    // (the state the turn started from is still in its chain's slot: had it changed, the turn would
    // have been squashed)
    auto decide = [&](const Turn &t, const double acc_lo, const double acc_hi) -> bool {
        double *slot = slots + t.q * sstride;
        double cur_c = slot[zs], best = slot[zs + 1];
        long long nacc = (long long)mc3_bits(slot[zs + 2]);
        const double xold = mine ? slot[lane] : 0.0;
        double xk = xold;
        bool accepted = false;
        if (t.go) {
            const double total = s_result[0];
            // (as WaveletHook::finish: the prior term of a problem without priors is 0.0)
            const double nextchisq = is_pos_inf(total) ? total : total + 0.0;
            if (!t.sjump) {
                accepted = metropolis_accept(0.5 * (cur_c - nextchisq), t.u, acc_lo, acc_hi);
            } else {                                   // chain.py:251-255
                const double dc = xold - t.zk, dn = t.npk - t.zk;
                const double cnorm = lane_serial_sum(mine ? dc * dc : 0.0, nf);
                const double nnorm = lane_serial_sum(mine ? dn * dn : 0.0, nf);
                const double mrfactor = mc3_det_pow_half(nnorm / cnorm, nf - 1);
                accepted = mc3_det_exp(0.5 * (cur_c - nextchisq)) * mrfactor > t.u;
            }
            if (accepted) {                            // chain.py:257-274
                xk = t.npk;
                cur_c = nextchisq;
                nacc += 1;
            }
        } else {
            oob += t.outk ? 1ull : 0ull;               // chain.py:242-243
        }
        const int lc = t.c - P.chain0;
        const long long row = M0 + t.R * nchains + t.c;               // chain.py:276-289
        if (mine)
            __hip_atomic_store(&Zr[(size_t)row * zs + lane], xk, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);             // publish (sc1)
        if (lane == 0) {
            log_post[row] = -0.5 * cur_c;
            zchain[row] = t.c;
        }
        if (accepted && cur_c < best) {
            best = cur_c;
            if (mine)
                bestx[(size_t)lc * nf + lane] = xk;
            if (lane == 0)
                best_iter[lc] = t.R;
        }
        if (mine)
            slot[lane] = xk;
        if (lane == 0) {
            slot[zs] = cur_c;
            slot[zs + 1] = best;
            slot[zs + 2] = mc3_from_bits((unsigned long long)nacc);
        }
        return accepted;
    };

    int par = 0, sn = 0;
    PIPE_DIAG_DECL;
    {
        jump_early();
        const int cmd0 = try_issue(np0);
        if (lane == 0)
            s_cmd[0] = cmd0;
        cur = cand;
    }
    PIPE_SYNC(0);                                      // (A): the first proposal (or the order)
    for (;;) {
        // (every later slot begins at barrier (1) of the one before: the proposal and the order
        // for it were written long before that)
        // (the order is this wave's own: read back, it is uniform without a branch on `cur`)
        const int cmd = __builtin_amdgcn_readfirstlane(s_cmd[par]);
        if (cmd == CW_STOP)
            break;
        // Level 0 of stage A is the long stretch (two compute waves per SIMD: ~3 700 cycles alone,
        // ~5 000 beside the helpers), and all of this wave's work goes into it.
        // While the tail wave is at the turn of slot s - 1: the turn after the one in flight -- its
        // draws, its rows -- and the logarithm the Metropolis test is decided with.
        PIPE_PART(-1);
        if (cur.valid) {
            const bool last_turn = qn + 1 == nturns;
            qn = last_turn ? 0 : qn + 1;
            Rn = last_turn ? Rn + 1 : Rn;
            tn += 1;
            cn = P.chain0 + b + qn * G;
            more = Rn < R0 + nref;
            if (more) {
                request_rows(draws_of(tn), Rn);
                if (need_normal)
                    normal_ahead = normal_draw((uint64_t)Rn, cn);
            }
        } else if (more) {
            request_rows(draws_of(tn), Rn);            // (a bubble: what was missing, again)
        }
        PIPE_PART(0);
        double acc_lo = 0.0, acc_hi = 0.0;
        if (pend.valid && pend.go) {
            accept_window(pend.u, acc_lo, acc_hi);
            asm volatile("; window" : "+v"(acc_lo), "+v"(acc_hi));
        }
        jump_early();
        if (pend.valid) {
            if (pend.go) {
                PIPE_PART(1);
                // the chisq: a word in LDS says so, long before the barrier does
                while (__hip_atomic_load(&s_cmd[2], __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP) != sn)
                    __builtin_amdgcn_s_sleep(1);
                PIPE_PART(2);
            }
            const bool accepted = decide(pend, acc_lo, acc_hi);
            if (accepted && cur.valid && cur.spec && cur.q == pend.q) {
                // the turn in flight started from a state that is no more: squashed, and what
                // was prepared behind it rewound to it
                cur.valid = false;
                Rn = cur.R;
                qn = cur.q;
                tn = cur.n;
                cn = cur.c;
                more = true;
                request_rows(draws_of(tn), Rn);
                if (need_normal)
                    normal_ahead = normal_draw((uint64_t)Rn, cn);
                jump_ready = false;
                rows_there = false;
            }
        }
        PIPE_PART(3);
        // the proposal for slot s + 1 (the other buffers: nobody reads those before (A))
        const int cmd_next = try_issue(par ? np0 : np1);
        PIPE_PART(4);
        if (lane == 0)
            s_cmd[par ^ 1] = cmd_next;
        PIPE_SYNC(1);                                  // the halo barriers of stage A
        PIPE_SYNC(2);
        PIPE_SYNC(3);
        if (nhalo > 3)
            PIPE_SYNC(4);
        PIPE_SYNC(5);                                  // (1) slot s
        pend = cur;
        cur = cand;
        par ^= 1;
        sn += 1;
#ifdef MC3HIP_PIPE_DIAG
        ++dg_n;
#endif
    }
    PIPE_DIAG_PRINT("chain");
#ifdef MC3HIP_PIPE_DIAG
    if (blockIdx.x == 0 && lane == 0)
        printf("chain wave: prepare %llu window %llu wait %llu decide %llu issue %llu\n",
               dg_part[0] / dg_n, dg_part[1] / dg_n, dg_part[2] / dg_n, dg_part[3] / dg_n,
               dg_part[4] / dg_n);
#endif
#ifdef MC3HIP_PIPE_DIAG
    if (blockIdx.x == 0 && lane == 0)
        printf("level 0 of waves 0..7: %llu %llu %llu %llu %llu %llu %llu %llu ns (%llu likelihoods)\n",
               mc3_pipe_diag[0] / mc3_pipe_diag[8], mc3_pipe_diag[1] / mc3_pipe_diag[8],
               mc3_pipe_diag[2] / mc3_pipe_diag[8], mc3_pipe_diag[3] / mc3_pipe_diag[8],
               mc3_pipe_diag[4] / mc3_pipe_diag[8], mc3_pipe_diag[5] / mc3_pipe_diag[8],
               mc3_pipe_diag[6] / mc3_pipe_diag[8], mc3_pipe_diag[7] / mc3_pipe_diag[8],
               mc3_pipe_diag[8]);
        printf("from the slot's start to level 0's, waves 0..7: %llu %llu %llu %llu %llu %llu %llu %llu ns\n",
               mc3_pipe_diag[24] / mc3_pipe_diag[8], mc3_pipe_diag[25] / mc3_pipe_diag[8],
               mc3_pipe_diag[26] / mc3_pipe_diag[8], mc3_pipe_diag[27] / mc3_pipe_diag[8],
               mc3_pipe_diag[28] / mc3_pipe_diag[8], mc3_pipe_diag[29] / mc3_pipe_diag[8],
               mc3_pipe_diag[30] / mc3_pipe_diag[8], mc3_pipe_diag[31] / mc3_pipe_diag[8]);
        printf("arrival at the first halo barrier, waves 0..7: %llu %llu %llu %llu %llu %llu %llu %llu ns\n",
               mc3_pipe_diag[32] / mc3_pipe_diag[8], mc3_pipe_diag[33] / mc3_pipe_diag[8],
               mc3_pipe_diag[34] / mc3_pipe_diag[8], mc3_pipe_diag[35] / mc3_pipe_diag[8],
               mc3_pipe_diag[36] / mc3_pipe_diag[8], mc3_pipe_diag[37] / mc3_pipe_diag[8],
               mc3_pipe_diag[38] / mc3_pipe_diag[8], mc3_pipe_diag[39] / mc3_pipe_diag[8]);
        printf("arrival at the second halo barrier, waves 0..7: %llu %llu %llu %llu %llu %llu %llu %llu ns\n",
               mc3_pipe_diag[40] / mc3_pipe_diag[8], mc3_pipe_diag[41] / mc3_pipe_diag[8],
               mc3_pipe_diag[42] / mc3_pipe_diag[8], mc3_pipe_diag[43] / mc3_pipe_diag[8],
               mc3_pipe_diag[44] / mc3_pipe_diag[8], mc3_pipe_diag[45] / mc3_pipe_diag[8],
               mc3_pipe_diag[46] / mc3_pipe_diag[8], mc3_pipe_diag[47] / mc3_pipe_diag[8]);
        printf("stage A up to its last barrier, waves 0..7: %llu %llu %llu %llu %llu %llu %llu %llu ns\n",
               mc3_pipe_diag[16] / mc3_pipe_diag[8], mc3_pipe_diag[17] / mc3_pipe_diag[8],
               mc3_pipe_diag[18] / mc3_pipe_diag[8], mc3_pipe_diag[19] / mc3_pipe_diag[8],
               mc3_pipe_diag[20] / mc3_pipe_diag[8], mc3_pipe_diag[21] / mc3_pipe_diag[8],
               mc3_pipe_diag[22] / mc3_pipe_diag[8], mc3_pipe_diag[23] / mc3_pipe_diag[8]);
#endif
    if (mine && oob)
        atomicAdd(&P.outbounds[lane], oob);
    for (int q = 0; q < nturns; ++q) {
        const int l = b + q * G;
        const double *slot = slots + q * sstride;
        if (mine)
            P.X[(size_t)l * nf + lane] = slot[lane];
        if (lane == 0) {
            P.chisq[l] = slot[zs];
            P.best_chisq[l] = slot[zs + 1];
            P.numaccept[l] = (long long)mc3_bits(slot[zs + 2]);
        }
    }
}

// ---- ONE chain per workgroup (C4): the chain wave of the pipelined kernel, again -------------
// pipe_chain_wave above serves any number of chains per workgroup and pays for it: the chains'
// states wait in LDS, every turn's draws are read out of a Philox batch word by word, its row
// addresses are 64-bit multiplications on one wave, the logarithm of its Metropolis draw is ~50
// dependent FP64 instructions -- ~5 000 cycles of ONE wave per slot (tools/pipe_diag.py:
// prepare 750, window 650, decide 1 500, issue 1 900), as long as level 0 of the compute waves
// beside it, and every vector instruction of it is taken from the compute waves of its SIMD.
// With ONE chain per workgroup (256 chains on 256 CUs: BASELINE config C4) the turn is the
// refresh and nothing of that is needed: this is the machinery of lean_snooker_body
// (kernels_lean.h) in the pipeline's slots --
//   * the state (xk, chisq, best, counters) lives in registers, what is the same in every lane
//     in SCALAR registers (said so once per slot), a turn's flags as bits of one word;
//   * everything a turn needs of its draws is worked out for 21 turns at once, a (turn, slot) per
//     lane: the offsets of its three history rows, the snooker-jump flag, gamma_s, the Metropolis
//     draw AND the window of its logarithm (handoff.h accept_window) -- a turn reads its values
//     out of the lanes (v_readlane), the logarithm has left the slot altogether;
//   * HISTORY ROWS THREE TURNS AHEAD.  A row is a random line of gigabytes of Z: ~1 200 cycles
//     from the request to the value, and pipe_chain_wave pays them every slot (asks at the top,
//     waits before the jump) and again after every squash.  Here the rows of turns T, T + 1 and
//     T + 2 are held (rv, rw, rx); a slot ENDS by asking for what the next one will lack -- one
//     turn's rows in the steady state -- so they travel through the barriers, and the rows the
//     turn in flight was proposed from are kept (cs_v, cs_w): a squash proposes again from
//     registers.  A row that was not written yet when it was asked for (the sentinel) is asked
//     for again where it is needed, as ever;
//   * RIDE: a proposal out of bounds is a reject whatever its likelihood would have been
//     (chain.py:239-243), and pipe_chain_wave gives it a slot of its own all the same -- the
//     compute waves idle through it (C4: 29 % of the turns).  Here such a turn rides with the one
//     behind it: the next turn is proposed at once, from the same state, into the same slot; the
//     rider's Z row and out-of-bounds count are written when the turn BEFORE it has been
//     decided (rejected: the state the rider started from stands; accepted: rider and turn are
//     squashed and proposed again).  One rider per turn;
//   * THE PROPOSAL BEFORE THE DECISION, as if the turn about to be decided were rejected (85 in
//     100 are): formed while the tail wave works, again after an accept;
//   * the protocol with the ten other waves (orders and proposals by slot parity, the chisq
//     word, the barriers) is pipe_chain_wave's.
// Same draws, same arithmetic in the same order: bit-identical (tests/test_gpu_multichain.py,
// the same cases as before).  Snooker without the support distribution (fepsilon = 0); the random
// walk and fepsilon != 0 keep pipe_chain_wave.  MC3HIP_NO_CW_ONE, MC3HIP_NO_CW_RIDE: the switches.
template <int SMP>
__device__ __forceinline__ void pipe_chain_wave_one(const DevProblem &P, const int nref,
                                                    double *np0, double *np1, int *s_cmd,
                                                    const double *s_result, const int nhalo)
{
    static_assert(SMP == MC3HIP_SNOOKER, "the history rows of a turn are what is prepared ahead");
    const int lane = threadIdx.x & 63;
    const int b = (int)blockIdx.x;
    const int c = P.chain0 + b;
    double *Zr = own_global(P.Z), *log_post = own_global(P.log_post);
    int32_t *zchain = own_global(P.zchain);
    const int nchains = P.nchains, nparams = P.nparams, nf = P.nfree, zs = P.zs;
    const long long M0 = P.M0;
    const double gamma = P.gamma;
    const bool mine = lane < nf;
    const int kk = mine ? lane : 0;
    const long long R0 = P.ctl[0];
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    double xk;
    {
        const double *snap_first = (R0 == 0) ? own_global(P.snap0)
                                             : Zr + (size_t)(M0 + (R0 - 1) * nchains) * zs;
        xk = mine ? snap_first[(size_t)c * zs + lane] : 0.0;
    }
    double cur_c = P.chisq[b], best = P.best_chisq[b];
    // (accepted turns of this launch as a 32-bit count, out-of-bounds trials 64-bit per lane:
    // two 64-bit "+= " in two branches are sunk into ONE store through a pointer that is the
    // one or the other -- and both counters then live in scratch)
    int nacc = 0;
    unsigned long long oob = 0;
    const int jfree = mine ? P.ifree[lane] : 0;
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    const long long abort_at = (P.test_abort >= 0 && b == 0) ? (long long)P.test_abort : -1;
    // the fixed parameters never change: written once (both buffers), the free ones every turn
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double *np = h ? np1 : np0;
        if (lane < nparams)
            np[lane] = P.params[lane];
        if (lane + 64 < nparams)
            np[lane + 64] = P.params[lane + 64];
    }
    // ---- the draws of NB turns at once: lane 3 i + s = slot s of turn bbase + i ----------------
    // A, B by slot: 0 -> the offsets (in doubles) of rows iR1, iR2; 1 -> the offset of row iz,
    // the jump flag; 2 -> gamma_s, u (as bits); blo / bhi: the window of log u (slot 2's lanes)
    constexpr int LPR = 3, NB = 21;
    unsigned long long A = 0, B = 0;
    double blo = 0.0, bhi = 0.0;
    int bbase = 0;                                      // (in turns since this launch began)
    const int bs = lane % LPR, bi = (lane / LPR < NB) ? lane / LPR : NB - 1;
    auto fill = [&](int n0) {                           // chain.py:195-205,259
        const long long Rl = R0 + n0 + bi;
        const Draw2 d = draw2(P.seed, (uint64_t)Rl, (uint32_t)c, (uint32_t)bs);
        const unsigned long long zsz = (unsigned long long)(M0 + Rl * nchains);
        if (bs == 0) {
            const unsigned long long iR1 = bounded(d.q0, zsz);
            unsigned long long iR2 = 1 + bounded(d.q1, zsz - 1);
            if (iR2 == iR1)
                iR2 = 0;
            A = iR1 * (unsigned long long)zs;
            B = iR2 * (unsigned long long)zs;
        } else if (bs == 1) {
            A = bounded(d.q1, zsz) * (unsigned long long)zs;
            B = (u01(d.q0) < 0.1) ? 1ull : 0ull;
        } else {
            A = mc3_bits(1.2 + (2.2 - 1.2) * u01(d.q0));
            B = mc3_bits(u01(d.q1));
        }
        accept_window(u01(d.q1), blo, bhi);             // (every lane: slot 2's keep theirs)
        bbase = n0;
    };
    auto lane64 = [&](unsigned long long v, int l) {
        return (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l) |
               ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l)
                << 32);
    };
    // (ONE place refills -- the top of a slot, below -- and it keeps the last three turns in the
    // batch: the turn about to be decided reads its Metropolis draw there, a squash rewinds by two)
    auto in_batch = [&](int n) { return LPR * (n - bbase); };
    // the history rows of turns tv, tv + 1, tv + 2; `have`: bit k = set k has been asked for
    double rv[3] = {}, rw[3] = {}, rx[3] = {};
    int tv = 0, have = 0;
    // ... and those the turn in flight was proposed from (its first turn's and the next one's)
    double cs_v[3] = {}, cs_w[3] = {};
    int cs_t = 0, cs_have = 0;
    // (the rows' addresses are formed again where a row is asked for again, not carried from
    // slot to slot: an array of pointers that lives across the loop ends up in scratch)
    auto row_addresses = [&](int n, const double *(&ra)[3]) {
        const int idx = in_batch(n);
        ra[0] = Zr + (lane64(A, idx) + (unsigned)kk);
        ra[1] = Zr + (lane64(B, idx) + (unsigned)kk);
        ra[2] = Zr + (lane64(A, idx + 1) + (unsigned)kk);
    };
    auto request = [&](int n, double (&r)[3]) {
        const double *ra[3];
        row_addresses(n, ra);
#pragma unroll
        for (int q = 0; q < 3; ++q)
            r[q] = load_published<false>(ra[q]);
    };
    // (what is the same in every lane as BITS of one word -- a flag of its own is a lane mask,
    // two scalar registers, and three turns of seven flags plus their Metropolis draws and
    // windows were ~60 scalar registers on top of the launch arguments: spilled to vector lanes,
    // every use a v_readlane.  The draw and its window are read out of the batch again when the
    // turn is decided.)
    struct Turn {
        int f;                                          // T_VALID | T_GO | ...
        int n;                                          // (its refresh: R0 + n)
        int lb;                                         // per lane: bit 0 outk, bit 1 router
        double npk, zk;
    };
    constexpr int T_VALID = 1, T_GO = 2, T_SJUMP = 4, T_SPEC = 8, T_RIDER = 16;
    Turn cur, pend, cand;                               // slot s, slot s - 1, slot s + 1
    cur.f = pend.f = cand.f = 0;
    cur.n = pend.n = cand.n = 0;
    cur.lb = pend.lb = cand.lb = 0;
    cur.npk = pend.npk = cand.npk = 0.0;
    cur.zk = pend.zk = cand.zk = 0.0;
    int tn = 0;                                         // the turn to issue next
    bool more = nref > 0;
    const bool ride_on = (P.cw_lean & 2) != 0 && nref > 1;
    double jump_ahead = 0.0, jump_second = 0.0;
    bool jump_ready = false, rows_there = false, second_ready = false, alive = true;

    // What is held of turn tn's rows (and of the next turn's, should this one ride) -- looked
    // at BEFORE the decision's stores: loads and stores retire through one counter, and a wait
    // behind the Z row's store waits for that store (~1 000 cycles).  The ordinary snooker jump
    // (nine in ten) at once.  tv == tn here.
    auto jump_early = [&]() {
        jump_ready = false;
        rows_there = false;
        second_ready = false;
        __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0)
        asm volatile("; rows" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]));
        asm volatile("; rows" : "+v"(rw[0]), "+v"(rw[1]), "+v"(rw[2]));
        asm volatile("; rows" : "+v"(rx[0]), "+v"(rx[1]), "+v"(rx[2]));
        if (more && (have & 1)) {
            const bool there2 = !__any(mc3_bits(rv[0]) == ROW_PENDING ||
                                       mc3_bits(rv[1]) == ROW_PENDING);
            rows_there = there2 && !__any(mc3_bits(rv[2]) == ROW_PENDING);
            const int idx = in_batch(tn);
            const bool sj = __builtin_amdgcn_readlane((int)(unsigned)B, idx + 1) != 0;
            if (there2 && !sj) {
                jump_ahead = gamma * (rv[0] - rv[1]) + P.fepsilon * 0.0;   // chain.py:217
                jump_ready = true;
            }
            // the turn behind it, should it ride: an ordinary jump whose two rows are there
            if (ride_on && (have & 2) && tn + 1 < nref) {
                const bool sj2 = __builtin_amdgcn_readlane((int)(unsigned)B, idx + LPR + 1) != 0;
                const bool there_w = !__any(mc3_bits(rw[0]) == ROW_PENDING ||
                                            mc3_bits(rw[1]) == ROW_PENDING);
                if (there_w && !sj2) {
                    jump_second = gamma * (rw[0] - rw[1]) + P.fepsilon * 0.0;
                    second_ready = true;
                }
            }
        }
    };
    // the proposal of turn tn into `np`; the order for the slot it opens
    auto try_issue = [&](double *np) -> int {
        cand.f = 0;
        if (!alive)
            return CW_STOP;
        if (!more)
            return (cur.f & T_VALID) ? CW_SKIP : CW_STOP;   // (the last turn's tail and decision)
        // SPECULATE ON REJECT (pipe_chain_wave): the chain's previous turn is the one in flight
        const bool spec = (cur.f & (T_VALID | T_GO)) == (T_VALID | T_GO);
        if (R0 + tn == abort_at) {                      // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            alive = false;
            return CW_STOP;
        }
        const int idx = in_batch(tn);
        double jump_k, zk = 0.0;
        bool sjump = false;
        if (jump_ready) {                               // chain.py:195-217
            jump_k = jump_ahead;
        } else {
            sjump = __builtin_amdgcn_readlane((int)(unsigned)B, idx + 1) != 0;
            if (!rows_there) {
                if (cur.f & T_VALID)
                    return CW_SKIP;                     // (never wait with a turn in flight)
                const double *ra[3];
                row_addresses(tn, ra);
                if (!(have & 1)) {
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        rv[q] = load_published<false>(ra[q]);
                    have |= 1;
                }
                alive = wait_rows<3, false>(ra, rv, ctl_rw) && alive;
                if (!alive)
                    return CW_STOP;
            }
            const double z1 = rv[0], z2 = rv[1];
            zk = rv[2];
            if (sjump) {
                const double gs = mc3_from_bits(lane64(A, idx + 2));
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
                jump_k = gamma * (z1 - z2) + P.fepsilon * 0.0;
            }
        }
        double npk = xk + jump_k;                       // chain.py:235-243
        bool outk = mine && (npk < pmin_k || npk > pmax_k);
        bool anyout = __any(outk);
        int f = T_VALID | (spec ? T_SPEC : 0), lb = 0;
        cand.n = tn;
        if (anyout && second_ready && R0 + tn + 1 != abort_at) {
            // RIDE: turn tn is rejected by the bounds; turn tn + 1 starts from the same state
            f |= T_RIDER;
            lb = outk ? 2 : 0;
            cand.n = tn + 1;
            sjump = false;
            zk = 0.0;
            npk = xk + jump_second;
            outk = mine && (npk < pmin_k || npk > pmax_k);
            anyout = __any(outk);
        }
        if (mine)
            np[jfree] = npk;
        cand.f = f | (anyout ? 0 : T_GO) | (sjump ? T_SJUMP : 0);
        cand.lb = lb | (outk ? 1 : 0);
        cand.npk = npk;
        cand.zk = zk;
        return anyout ? CW_SKIP : CW_GO;
    };
    // the Z row and the books of the turn that rides with `t` -- rejected by the bounds, the
    // state it started from stands (chain.py:239-243,276-289)
    auto settle_rider = [&](const Turn &t) {
        oob += (t.lb & 2) ? 1ull : 0ull;
        const long long row = M0 + (R0 + t.n - 1) * nchains + c;
        if (mine)
            __hip_atomic_store(&Zr[(size_t)row * zs + lane], xk, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);             // publish (sc1)
        if (lane == 0) {
            log_post[row] = -0.5 * cur_c;
            zchain[row] = c;
        }
    };
    // Metropolis, Z append and the books of a turn whose chisq is in s_result (the state the
    // turn started from is still xk: had it changed, the turn would have been squashed)
    auto decide = [&](const Turn &t) -> bool {
        const double xold = xk;
        bool accepted = false;
        if (t.f & T_GO) {
            // (the same word in every lane -- SAID so: a value read from LDS is a per-lane value
            // to the compiler, the Metropolis test on it a per-lane branch, and everything the
            // squash behind it assigns would be kept per lane from then on)
            const double total = uniform_double(s_result[0]);
            // (as WaveletHook::finish: the prior term of a problem without priors is 0.0)
            const double nextchisq = is_pos_inf(total) ? total : total + 0.0;
            // the turn's Metropolis draw and the window of its logarithm: still in the batch
            const int at = in_batch(t.n) + 2;
            const double u = mc3_from_bits(lane64(B, at));            // chain.py:259
            if (__builtin_expect(!(t.f & T_SJUMP), 1)) {
                accepted = metropolis_accept(0.5 * (cur_c - nextchisq), u, read_lane(blo, at),
                                             read_lane(bhi, at));
            } else {                                   // chain.py:251-255
                const double dc = xold - t.zk, dn = t.npk - t.zk;
                const double cnorm = lane_serial_sum(mine ? dc * dc : 0.0, nf);
                const double nnorm = lane_serial_sum(mine ? dn * dn : 0.0, nf);
                const double mrfactor = mc3_det_pow_half(nnorm / cnorm, nf - 1);
                accepted = mc3_det_exp(0.5 * (cur_c - nextchisq)) * mrfactor > u;
            }
            if (accepted) {                            // chain.py:257-274
                xk = t.npk;
                cur_c = nextchisq;
                nacc += 1;
            }
        } else {
            oob += (t.lb & 1) ? 1ull : 0ull;           // chain.py:242-243
        }
        const long long row = M0 + (R0 + t.n) * nchains + c;          // chain.py:276-289
        if (mine)
            __hip_atomic_store(&Zr[(size_t)row * zs + lane], xk, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);             // publish (sc1)
        if (lane == 0) {
            log_post[row] = -0.5 * cur_c;
            zchain[row] = c;
        }
        if (__builtin_expect(accepted && cur_c < best, 0)) {
            best = cur_c;
            if (mine)
                P.bestx[(size_t)b * nf + lane] = xk;
            if (lane == 0)
                P.best_iter[b] = R0 + t.n;
        }
        return accepted;
    };
    // The end of a slot's work: the rows the NEXT slot proposes from.  Its turn is the one behind
    // `cand` (or this one again: a bubble); what is held of it moves up, what is not is asked for
    // now -- and travels through the barriers.  Before that, the rows `cand` was proposed from are
    // put aside: should it be squashed, it is proposed again from them.
    auto rows_ahead = [&]() {
        const int t_next = (cand.f & T_VALID) ? cand.n + 1 : tn;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            cs_v[q] = rv[q];
            cs_w[q] = rw[q];
        }
        cs_t = tv;
        cs_have = have & 3;
        const int d = t_next - tv;
        if (d == 1) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                rv[q] = rw[q];
                rw[q] = rx[q];
            }
            have >>= 1;
        } else if (d == 2) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
                rv[q] = rx[q];
            have >>= 2;
        } else if (d == 0) {
            // (a bubble: rows of this turn were not there -- asked for again)
            if (!rows_there)
                have &= ~1;
        } else {
            have = 0;
        }
        tv = t_next;
        if (!(have & 1) && tv < nref)
            request(tv, rv);
        if (!(have & 2) && tv + 1 < nref)
            request(tv + 1, rw);
        if (!(have & 4) && tv + 2 < nref)
            request(tv + 2, rx);
        have = (tv < nref ? 1 : 0) | (tv + 1 < nref ? 2 : 0) | (tv + 2 < nref ? 4 : 0);
    };

    int par = 0, sn = 0;
    PIPE_DIAG_DECL;
    fill(0);
    {
        request(0, rv);
        have = 1;
        if (nref > 1) {
            request(1, rw);
            have |= 2;
        }
        if (nref > 2) {
            request(2, rx);
            have |= 4;
        }
        jump_early();
        const int cmd0 = try_issue(np0);
        if (lane == 0)
            s_cmd[0] = cmd0;
        rows_ahead();
        cur = cand;
    }
    PIPE_SYNC(0);                                      // (A): the first proposal (or the order)
    for (;;) {
        // (the order is this wave's own: read back, it is uniform without a branch on `cur`)
        const int cmd = __builtin_amdgcn_readfirstlane(s_cmd[par]);
        if (cmd == CW_STOP)
            break;
        // The control state of the pipeline is the same in every lane -- SAID so once per slot:
        // left to itself the compiler keeps the turn counter in vector registers and the turns'
        // flags as lane masks (every `if` a save / restore of EXEC, every batch index a
        // v_readfirstlane in front of its v_readlane).
        {
            auto ub = [](bool v) { return __builtin_amdgcn_readfirstlane(v ? 1 : 0) != 0; };
            tn = __builtin_amdgcn_readfirstlane(tn);
            tv = __builtin_amdgcn_readfirstlane(tv);
            have = __builtin_amdgcn_readfirstlane(have);
            cs_t = __builtin_amdgcn_readfirstlane(cs_t);
            cs_have = __builtin_amdgcn_readfirstlane(cs_have);
            bbase = __builtin_amdgcn_readfirstlane(bbase);
            more = ub(more);
            alive = ub(alive);
            cur.f = __builtin_amdgcn_readfirstlane(cur.f);
            cur.n = __builtin_amdgcn_readfirstlane(cur.n);
            pend.f = __builtin_amdgcn_readfirstlane(pend.f);
            pend.n = __builtin_amdgcn_readfirstlane(pend.n);
            cur_c = uniform_double(cur_c);
            best = uniform_double(best);
        }
        PIPE_PART(-1);
        // the turn after the one in flight (tv is there already: rows_ahead)
        if (cur.f & T_VALID) {
            tn = cur.n + 1;
            more = tn < nref;
            // (the batch holds turns tn - 3 .. tn + 4: the turn about to be decided reads its
            // Metropolis draw there, a squash rewinds by two at most, rows are asked for up to
            // four turns ahead)
            if (__builtin_expect(tn + 4 >= bbase + NB, 0))
                fill(tn >= 3 ? tn - 3 : 0);
        }
        PIPE_PART(0);
        jump_early();
        // THE PROPOSAL FIRST, as if the turn about to be decided were rejected.  Never where
        // forming it could wait for rows: the turn to be decided is still in flight.
        const int buf = par ^ 1;
        bool formed = false;
        int cmd_next = CW_SKIP;
        if (alive && more && (jump_ready || rows_there) && R0 + tn != abort_at) {
            cmd_next = try_issue(buf ? np1 : np0);
            formed = true;
        }
        PIPE_PART(1);
        if (pend.f & T_VALID) {
            if (pend.f & T_GO) {
                // the chisq: a word in LDS says so, long before the barrier does
                // (the word is the same in every lane, and said so: a per-lane exit of this
                // loop is a per-lane exit in the compiler's eyes)
                while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(
                           &s_cmd[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != sn)
                    __builtin_amdgcn_s_sleep(1);
            }
            PIPE_PART(2);
            const bool accepted = decide(pend);
            if (accepted && (cur.f & (T_VALID | T_SPEC)) == (T_VALID | T_SPEC)) {
                // the turn in flight (and its rider) started from a state that is no more:
                // squashed, and proposed again -- from the rows it was proposed from
                cur.f = 0;
                tn = cs_t;
                more = true;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    rv[q] = cs_v[q];
                    rw[q] = cs_w[q];
                }
                tv = cs_t;
                have = cs_have;
                jump_early();
            }
            if (accepted)
                formed = false;                         // (the state has changed: again)
        } else {
            PIPE_PART(2);
        }
        // the turn before the one in flight is settled: so is the rider between them
        if ((cur.f & (T_VALID | T_RIDER)) == (T_VALID | T_RIDER))
            settle_rider(cur);
        PIPE_PART(3);
        // the proposal for slot s + 1 (the other buffer: nobody reads it before (A)), unless it
        // stands already
        if (!formed)
            cmd_next = try_issue(buf ? np1 : np0);
        if (lane == 0)
            s_cmd[buf] = cmd_next;
        rows_ahead();
        PIPE_PART(4);
        PIPE_SYNC(1);                                  // the halo barriers of stage A
        PIPE_SYNC(2);
        PIPE_SYNC(3);
        if (nhalo > 3)
            PIPE_SYNC(4);
        PIPE_SYNC(5);                                  // (1) slot s
        pend = cur;
        cur = cand;
        par ^= 1;
        sn += 1;
#ifdef MC3HIP_PIPE_DIAG
        ++dg_n;
#endif
    }
    PIPE_DIAG_PRINT("chain");
#ifdef MC3HIP_PIPE_DIAG
    if (blockIdx.x == 0 && lane == 0) {
        printf("chain wave (one chain): prepare %llu window %llu wait %llu decide %llu issue %llu\n",
               dg_part[0] / dg_n, dg_part[1] / dg_n, dg_part[2] / dg_n, dg_part[3] / dg_n,
               dg_part[4] / dg_n);
        printf("level 0 of waves 0..7: %llu %llu %llu %llu %llu %llu %llu %llu ns (%llu likelihoods)\n",
               mc3_pipe_diag[0] / mc3_pipe_diag[8], mc3_pipe_diag[1] / mc3_pipe_diag[8],
               mc3_pipe_diag[2] / mc3_pipe_diag[8], mc3_pipe_diag[3] / mc3_pipe_diag[8],
               mc3_pipe_diag[4] / mc3_pipe_diag[8], mc3_pipe_diag[5] / mc3_pipe_diag[8],
               mc3_pipe_diag[6] / mc3_pipe_diag[8], mc3_pipe_diag[7] / mc3_pipe_diag[8],
               mc3_pipe_diag[8]);
        printf("from the slot's start to level 0's, waves 0..7: %llu %llu %llu %llu %llu %llu %llu %llu ns\n",
               mc3_pipe_diag[24] / mc3_pipe_diag[8], mc3_pipe_diag[25] / mc3_pipe_diag[8],
               mc3_pipe_diag[26] / mc3_pipe_diag[8], mc3_pipe_diag[27] / mc3_pipe_diag[8],
               mc3_pipe_diag[28] / mc3_pipe_diag[8], mc3_pipe_diag[29] / mc3_pipe_diag[8],
               mc3_pipe_diag[30] / mc3_pipe_diag[8], mc3_pipe_diag[31] / mc3_pipe_diag[8]);
        printf("arrival at the first halo barrier, waves 0..7: %llu %llu %llu %llu %llu %llu %llu %llu ns\n",
               mc3_pipe_diag[32] / mc3_pipe_diag[8], mc3_pipe_diag[33] / mc3_pipe_diag[8],
               mc3_pipe_diag[34] / mc3_pipe_diag[8], mc3_pipe_diag[35] / mc3_pipe_diag[8],
               mc3_pipe_diag[36] / mc3_pipe_diag[8], mc3_pipe_diag[37] / mc3_pipe_diag[8],
               mc3_pipe_diag[38] / mc3_pipe_diag[8], mc3_pipe_diag[39] / mc3_pipe_diag[8]);
        printf("arrival at the second halo barrier, waves 0..7: %llu %llu %llu %llu %llu %llu %llu %llu ns\n",
               mc3_pipe_diag[40] / mc3_pipe_diag[8], mc3_pipe_diag[41] / mc3_pipe_diag[8],
               mc3_pipe_diag[42] / mc3_pipe_diag[8], mc3_pipe_diag[43] / mc3_pipe_diag[8],
               mc3_pipe_diag[44] / mc3_pipe_diag[8], mc3_pipe_diag[45] / mc3_pipe_diag[8],
               mc3_pipe_diag[46] / mc3_pipe_diag[8], mc3_pipe_diag[47] / mc3_pipe_diag[8]);
        printf("stage A up to its last barrier, waves 0..7: %llu %llu %llu %llu %llu %llu %llu %llu ns\n",
               mc3_pipe_diag[16] / mc3_pipe_diag[8], mc3_pipe_diag[17] / mc3_pipe_diag[8],
               mc3_pipe_diag[18] / mc3_pipe_diag[8], mc3_pipe_diag[19] / mc3_pipe_diag[8],
               mc3_pipe_diag[20] / mc3_pipe_diag[8], mc3_pipe_diag[21] / mc3_pipe_diag[8],
               mc3_pipe_diag[22] / mc3_pipe_diag[8], mc3_pipe_diag[23] / mc3_pipe_diag[8]);
    }
#endif
    if (mine && oob)
        atomicAdd(&P.outbounds[lane], oob);
    if (mine)
        P.X[(size_t)b * nf + lane] = xk;
    if (lane == 0) {
        P.chisq[b] = cur_c;
        P.best_chisq[b] = best;
        P.numaccept[b] += (long long)nacc;
    }
}

template <int MT, int SMP, bool ONE = false>
__global__ __launch_bounds__(704) void wavelet_pipe_kernel(const DevProblem P, const int nref)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SmemLayout L = carve(smem, P, 512);           // the compute waves' areas
    double *own = L.xs + (P.ndata > 128 ? P.ndata : 128);
    // (the area the general chain would use behind the nine-wave kernel: the second proposal and
    // the second noise table)
    double *np1 = own, *tab1 = own + MC3HIP_MAX_PARAMS;
    static_assert(MC3HIP_MAX_PARAMS + 48 <= CW_OWN, "room behind the compute waves' areas");
    int *s_cmd = reinterpret_cast<int *>(own + CW_OWN);
    double *s_result = own + CW_OWN + 2;
    const int tid = threadIdx.x;
    if (launch_given_up(P.ctl, s_cmd + 3))             // (one decision per workgroup)
        return;
    if (tid < 512)
        wavelet_stage<8, MT>(P, L.xs);
    block_sync<9>();
    const int B = P.ndata / 512;                        // 16 or 32 (mc3hip.hip choose_wpc)
    const int nhalo = ilog2(B) - 1;                     // 3 or 4 (the waves below count on it)
    const int M = ilog2(P.ndata);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (the chain wave and the tail wave are what a slot waits for; the noise wave has time)
    // (MC3HIP_CW_PRIO / MC3HIP_TAIL_PRIO: tools/variant_lib.py builds the library with other
    // priorities for the two -- measurements only)
#ifndef MC3HIP_CW_PRIO
#define MC3HIP_CW_PRIO 3
#endif
#ifndef MC3HIP_TAIL_PRIO
#define MC3HIP_TAIL_PRIO 3
#endif
#if MC3HIP_CW_PRIO == MC3HIP_TAIL_PRIO
    if (wave == 8 || wave == 9)
        __builtin_amdgcn_s_setprio(MC3HIP_CW_PRIO);
#else
    if (wave == 8)
        __builtin_amdgcn_s_setprio(MC3HIP_CW_PRIO);
    if (wave == 9)
        __builtin_amdgcn_s_setprio(MC3HIP_TAIL_PRIO);
#endif
    if (wave == 8) {
        if constexpr (ONE)
            pipe_chain_wave_one<SMP>(P, nref, L.nextp, np1, s_cmd, s_result, nhalo);
        else
            pipe_chain_wave<SMP>(P, nref, L.nextp, np1, L.tab, tab1, s_cmd, s_result, nhalo, M,
                                 smem + P.mc_off);
        return;
    }
    if (wave == 10) {
        // ---- the noise wave: the noise terms of slot s's proposal (software exp2 / log /
        // reciprocals, the ordered log-determinant sum: ~2 500 dependent cycles), needed by the
        // end of stage A; a wave of its own so that the three helpers sit on three SIMDs
        int par = 0;
        PIPE_DIAG_DECL;
        PIPE_SYNC(0);                                   // (A)
        for (;;) {
            const int cmd = s_cmd[par];
            if (cmd == CW_STOP)
                break;
            if (cmd == CW_GO)
                wavelet_noise_terms(par ? np1 : L.nextp, P.nparams, M, par ? tab1 : L.tab);
            PIPE_SYNC(1);                               // the halo barriers of stage A
            PIPE_SYNC(2);
            PIPE_SYNC(3);
            if (nhalo > 3)
                PIPE_SYNC(4);
            PIPE_SYNC(5);                               // (1) slot s
            par ^= 1;
#ifdef MC3HIP_PIPE_DIAG
            ++dg_n;
#endif
        }
        PIPE_DIAG_PRINT("noise");
        return;
    }
    if (wave == 9) {
        // ---- the tail wave ------------------------------------------------------------------
        PipeTail<8> tail;
        bool have = false;
        int par = 0, sn = 0;
        if ((tid & 63) == 0)
            s_cmd[2] = -1;
        PIPE_DIAG_DECL;
        PIPE_SYNC(0);                                   // (A)
        for (;;) {
            const int cmd = s_cmd[par];
            if (cmd == CW_STOP)
                break;
            if (have) {
                // the turn of slot s - 1 (its noise table and its flag are the other ones), all of
                // it before the first halo barrier: level 0 takes the compute waves longer than
                // this, and what is read here they overwrite at the END of stage A
                tail.load(L.part, L.halo, L.flag + 2 * (par ^ 1));
                const double *tab = par ? L.tab : tab1;
                const double c = (B == 32) ? tail.run<32>(tab, M) : tail.run<16>(tab, M);
                if ((tid & 63) == 0) {
                    s_result[0] = c;
                    __hip_atomic_store(&s_cmd[2], sn, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);     // (LDS keeps the order)
                }
            }
            PIPE_SYNC(1);                               // the halo barriers of stage A
            PIPE_SYNC(2);
            PIPE_SYNC(3);
            if (nhalo > 3)
                PIPE_SYNC(4);
            PIPE_SYNC(5);                               // (1) slot s
            have = cmd == CW_GO;
            par ^= 1;
            sn += 1;
#ifdef MC3HIP_PIPE_DIAG
            ++dg_n;
#endif
        }
        PIPE_DIAG_PRINT("tail");
        return;
    }
    // ---- the compute waves -------------------------------------------------------------------
    int par = 0;
    block_sync<9>();                                    // (A)
    for (;;) {
#ifdef MC3HIP_PIPE_DIAG
        const unsigned long long dg_slot0 = 10ull * __builtin_amdgcn_s_memrealtime();
#define PIPE_DIAG_SLOT0 , dg_slot0
#else
#define PIPE_DIAG_SLOT0
#endif
        const int cmd = s_cmd[par];
        const double *np = par ? np1 : L.nextp;
        double *tab = par ? tab1 : L.tab;
        int *flag = L.flag + 2 * par;                   // ("a model value was +inf": flag[1])
        Model<MT> model;
        model.init(P, np);
        if (cmd == CW_STOP)
            return;
        par ^= 1;
        if (cmd != CW_GO) {
            for (int k = 0; k <= nhalo; ++k)
                block_sync<9>();                        // (the slot belongs to the helper waves)
            continue;
        }
        if (B == 32)
            wavelet_body<8, MT, 32, false, true, true>(P, np, L.part, flag, tab, L.halo, L.xs,
                                                       nullptr, 0, nullptr, 0, &model PIPE_DIAG_SLOT0);
        else
            wavelet_body<8, MT, 16, false, true, true>(P, np, L.part, flag, tab, L.halo, L.xs,
                                                       nullptr, 0, nullptr, 0, &model PIPE_DIAG_SLOT0);
#undef PIPE_DIAG_SLOT0
    }
}

// The helper wave of the one-wave kernels below (wave 1 of a workgroup of two): the noise
// terms of every proposal, while wave 0 transforms.  Barrier (A): a proposal is in np -- or
// (flag[3]) the launch is over; barrier (B): wavelet_body<1, .., NX>, where the terms are used.
__device__ __forceinline__ void wavelet_one_helper(const DevProblem &P, const SmemLayout &L)
{
    const int M = ilog2(P.ndata);
    for (;;) {
        block_sync<2>();                                // (A)
        if (L.flag[3] != 0)
            return;
        wavelet_noise_terms(L.nextp, P.nparams, M, L.tab);
        block_sync<2>();                                // (B)
    }
}

// 4 096 points -- four waves per chain -- in the same kernels (WPC = 4): the chain is wave 0's,
// waves 1..3 take part in every likelihood (the noise terms are the last wave's, behind its
// share of stage A: wavelet_body) and in nothing else.  Barrier (A) as above.
template <int MT, int WPC>
__device__ __forceinline__ void wavelet_one_followers(const DevProblem &P, const SmemLayout &L)
{
    if constexpr (WPC == 1) {
        wavelet_one_helper(P, L);
    } else {
        for (;;) {
            block_sync<WPC>();                          // (A)
            if (L.flag[3] != 0)
                return;
            (void)block_wavelet<WPC, MT, false>(P, L.nextp, L.part, L.flag, L.tab, L.halo, L.xs);
        }
    }
}
// wave 0: the likelihood of the proposal in L.nextp
template <int MT, int WPC>
__device__ __forceinline__ double wavelet_one_likelihood(const DevProblem &P, const SmemLayout &L)
{
    if constexpr (WPC == 1) {
        block_sync<2>();                                // (A) the helper wave's turn begins
        return block_wavelet<1, MT, false, true>(P, L.nextp, L.part, L.flag, L.tab, L.halo, L.xs);
    } else {
        block_sync<WPC>();                              // (A) everybody's
        return block_wavelet<WPC, MT, false>(P, L.nextp, L.part, L.flag, L.tab, L.halo, L.xs);
    }
}
// wave 0, the launch over: the other waves leave
template <int WPC>
__device__ __forceinline__ void wavelet_one_leave(const SmemLayout &L)
{
    if ((threadIdx.x & 63) == 0)
        L.flag[3] = 1;
    if constexpr (WPC == 1)
        block_sync<2>();
    else
        block_sync<WPC>();
}

// ---- ONE wave per chain: wavelet likelihoods of up to 2 048 points, by hand ------------------
// Below 8 192 points there was no chain wave: the general chain (run_chain behind
// persist_kernel) ran these shapes -- 3.8 us per iteration at 64 points, 5.6 at 2 048, of which
// the likelihood itself is the smaller part (tools/mid_range.py).  This is pipe_chain_wave_one's
// bookkeeping -- state in registers, a turn's draws / row offsets / Metropolis window 21 turns at
// a time, history rows asked for a turn ahead (they travel while the likelihood runs) -- around
// wavelet_body<1>: propose, bounds, transform, decide, publish, one turn after the other.  Snooker
// (the support distribution's deviates, if any, drawn in the turn), the shapes of host::cw_lean_eligible (thinning 1, no trace,
// no priors, no shared parameters, strict lockstep), one chain per workgroup, one GPU.  Same
// draws, same arithmetic in the same order as run_chain: bit-identical
// (tests/test_gpu_parity.py).  MC3HIP_NO_WL_ONE: the general chain.
template <int MT, int WPC = 1>
__global__ __launch_bounds__(WPC == 1 ? 128 : 64 * WPC) void wavelet_one_kernel(const DevProblem P,
                                                                               const int nref)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SmemLayout L = carve(smem, P, 64 * WPC);
    const int lane = threadIdx.x & 63;
    const int b = (int)blockIdx.x;
    const int c = P.chain0 + b;
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    // (one decision per workgroup, in the word that tells the follower waves the launch is
    // over: none of them can be left behind a wave that saw another value)
    if (launch_given_up(ctl_rw, &L.flag[3]))
        return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if constexpr (WPC > 1)
        wavelet_stage<WPC, MT>(P, L.xs);                // (everybody's share of the abscissae)
    // (one wave: wave 1 has the noise terms, wavelet_one_helper; four: waves 1..3 their share
    // of every likelihood)
    if (wave != 0) {
        wavelet_one_followers<MT, WPC>(P, L);
        return;
    }
    if (lane == 0)
        L.flag[3] = 0;
    if constexpr (WPC == 1)
        wavelet_stage<1, MT>(P, L.xs);
    double *np = L.nextp;
    if (lane < P.nparams)
        np[lane] = P.params[lane];
    if (lane + 64 < P.nparams)
        np[lane + 64] = P.params[lane + 64];
    block_sync<1>();
    double *Zr = own_global(P.Z), *log_post = own_global(P.log_post);
    int32_t *zchain = own_global(P.zchain);
    const int nchains = P.nchains, nf = P.nfree, zs = P.zs;
    const long long M0 = P.M0;
    const double gamma = P.gamma;
    const bool mine = lane < nf;
    const int kk = mine ? lane : 0;
    const long long R0 = P.ctl[0];
    double xk;
    {
        const double *snap_first = (R0 == 0) ? own_global(P.snap0)
                                             : Zr + (size_t)(M0 + (R0 - 1) * nchains) * zs;
        xk = mine ? snap_first[(size_t)c * zs + lane] : 0.0;
    }
    double cur_c = P.chisq[b], best = P.best_chisq[b];
    int nacc = 0;
    unsigned long long oob = 0;
    const int jfree = mine ? P.ifree[lane] : 0;
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    const long long abort_at = (P.test_abort >= 0 && b == 0) ? (long long)P.test_abort : -1;
    // (the support distribution, chain.py:185,217: a normal deviate per free parameter and turn)
    const bool need_normal = P.fepsilon != 0.0;
    const double pstep_k = (mine && need_normal) ? P.fstep[lane] : 0.0;
    auto normal_draw = [&](uint64_t it) {              // as run_chain's
        double nk = 0.0;
        if (mine) {
            const Draw2 d = draw2(P.seed, it, (uint32_t)c, (uint32_t)(3 + (lane >> 1)));
            const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(d.q0)));
            double sn, cs;
            mc3_det_sincos2pi(u01(d.q1), &sn, &cs);
            nk = (rad * ((lane & 1) ? sn : cs)) * pstep_k;
        }
        return nk;
    };
    // the draws of NB turns at once (pipe_chain_wave_one): lane 3 i + s = slot s of turn bbase + i
    constexpr int LPR = 3, NB = 21;
    unsigned long long A = 0, B = 0;
    double blo = 0.0, bhi = 0.0;
    int bbase = 0;
    const int bs = lane % LPR, bi = (lane / LPR < NB) ? lane / LPR : NB - 1;
    auto fill = [&](int n0) {                           // chain.py:195-205,259
        const long long Rl = R0 + n0 + bi;
        const Draw2 d = draw2(P.seed, (uint64_t)Rl, (uint32_t)c, (uint32_t)bs);
        const unsigned long long zsz = (unsigned long long)(M0 + Rl * nchains);
        if (bs == 0) {
            const unsigned long long iR1 = bounded(d.q0, zsz);
            unsigned long long iR2 = 1 + bounded(d.q1, zsz - 1);
            if (iR2 == iR1)
                iR2 = 0;
            A = iR1 * (unsigned long long)zs;
            B = iR2 * (unsigned long long)zs;
        } else if (bs == 1) {
            A = bounded(d.q1, zsz) * (unsigned long long)zs;
            B = (u01(d.q0) < 0.1) ? 1ull : 0ull;
        } else {
            A = mc3_bits(1.2 + (2.2 - 1.2) * u01(d.q0));
            B = mc3_bits(u01(d.q1));
        }
        accept_window(u01(d.q1), blo, bhi);
        bbase = n0;
    };
    auto lane64 = [&](unsigned long long v, int l) {
        return (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l) |
               ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l)
                << 32);
    };
    auto row_addresses = [&](int n, const double *(&ra)[3]) {
        const int idx = LPR * (n - bbase);
        ra[0] = Zr + (lane64(A, idx) + (unsigned)kk);
        ra[1] = Zr + (lane64(B, idx) + (unsigned)kk);
        ra[2] = Zr + (lane64(A, idx + 1) + (unsigned)kk);
    };
    double rv[3] = {}, rw[3] = {};
    fill(0);
    if (nref > 0) {
        const double *ra[3];
        row_addresses(0, ra);
#pragma unroll
        for (int q = 0; q < 3; ++q)
            rv[q] = load_published<false>(ra[q]);
    }
    int done = 0;
    for (int n = 0; n < nref; ++n) {
        if (__builtin_expect(R0 + n == abort_at, 0)) {                 // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        // this turn's rows: asked for a turn ago (a row that was not written yet: asked again)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("; rows" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]));
        int idx = LPR * (n - bbase);
        const bool sjump = __builtin_amdgcn_readlane((int)(unsigned)B, idx + 1) != 0;
        {
            const bool there2 = !__any(mc3_bits(rv[0]) == ROW_PENDING ||
                                       mc3_bits(rv[1]) == ROW_PENDING);
            const bool there3 = there2 && !__any(mc3_bits(rv[2]) == ROW_PENDING);
            if (__builtin_expect(!(sjump ? there3 : there2), 0)) {
                const double *ra[3];
                row_addresses(n, ra);
                if (!wait_rows<3, false>(ra, rv, ctl_rw))
                    break;
            }
        }
        const double z1 = rv[0], z2 = rv[1], zk = rv[2];
        const double gs = mc3_from_bits(lane64(A, idx + 2));
        const double u = mc3_from_bits(lane64(B, idx + 2));           // chain.py:259
        const double lo = read_lane(blo, idx + 2), hi = read_lane(bhi, idx + 2);
        // the next turn's rows travel while this one's likelihood runs
        if (n + 1 < nref) {
            if (__builtin_expect(n + 1 >= bbase + NB, 0))
                fill(n + 1);
            const double *ra[3];
            row_addresses(n + 1, ra);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                rw[q] = load_published<false>(ra[q]);
        }
        double jump_k;
        if (__builtin_expect(sjump, 0)) {                             // chain.py:207-215
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
            const double normal_k = need_normal ? normal_draw((uint64_t)(R0 + n)) : 0.0;
            jump_k = gamma * (z1 - z2) + P.fepsilon * normal_k;        // chain.py:217
        }
        const double npk = xk + jump_k;                               // chain.py:235-243
        const bool outk = mine && (npk < pmin_k || npk > pmax_k);
        bool accepted = false;
        if (!__any(outk)) {
            if (mine)
                np[jfree] = npk;
            const double total = wavelet_one_likelihood<MT, WPC>(P, L);
            // (as WaveletHook::finish: the prior term of a problem without priors is 0.0)
            const double nextchisq = is_pos_inf(total) ? total : total + 0.0;
            if (__builtin_expect(!sjump, 1)) {
                accepted = metropolis_accept(0.5 * (cur_c - nextchisq), u, lo, hi);
            } else {                                                  // chain.py:251-255
                const double dc = xk - zk, dn = npk - zk;
                const double cnorm = lane_serial_sum(mine ? dc * dc : 0.0, nf);
                const double nnorm = lane_serial_sum(mine ? dn * dn : 0.0, nf);
                const double mrfactor = mc3_det_pow_half(nnorm / cnorm, nf - 1);
                accepted = mc3_det_exp(0.5 * (cur_c - nextchisq)) * mrfactor > u;
            }
            if (accepted) {                                           // chain.py:257-274
                xk = npk;
                cur_c = nextchisq;
                nacc += 1;
            }
        } else {
            oob += outk ? 1ull : 0ull;                                // chain.py:242-243
        }
        const long long row = M0 + (R0 + n) * nchains + c;            // chain.py:276-289
        if (mine)
            __hip_atomic_store(&Zr[(size_t)row * zs + lane], xk, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);             // publish (sc1)
        if (lane == 0) {
            log_post[row] = -0.5 * cur_c;
            zchain[row] = c;
        }
        if (__builtin_expect(accepted && cur_c < best, 0)) {
            best = cur_c;
            if (mine)
                P.bestx[(size_t)b * nf + lane] = xk;
            if (lane == 0)
                P.best_iter[b] = R0 + n;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
            rv[q] = rw[q];
        done = n + 1;
    }
    (void)done;
    wavelet_one_leave<WPC>(L);
    if (mine && oob)
        atomicAdd(&P.outbounds[lane], oob);
    if (mine)
        P.X[(size_t)b * nf + lane] = xk;
    if (lane == 0) {
        P.chisq[b] = cur_c;
        P.best_chisq[b] = best;
        P.numaccept[b] += (long long)nacc;
    }
}

// The same for the two samplers that need no history: the random walk (chain.py:219-220 -- no
// rows at all, a normal draw per free parameter) and DEMC (chain.py:221-232 -- the two partners' rows of the refresh before, asked for right behind the
// own row).  The draws of a turn are one scalar Philox call, made a turn ahead like the normal
// draws (they run in the shadow of the partners' rows).  Same shapes as above.
template <int MT, int SMP, int WPC = 1>
__global__ __launch_bounds__(WPC == 1 ? 128 : 64 * WPC) void wavelet_one_other_kernel(
    const DevProblem P, const int nref)
{
    static_assert(SMP == MC3HIP_MRW || SMP == MC3HIP_DEMC, "");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SmemLayout L = carve(smem, P, 64 * WPC);
    const int lane = threadIdx.x & 63;
    const int b = (int)blockIdx.x;
    const int c = P.chain0 + b;
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    // (one decision per workgroup, in the word that tells the follower waves the launch is
    // over: none of them can be left behind a wave that saw another value)
    if (launch_given_up(ctl_rw, &L.flag[3]))
        return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if constexpr (WPC > 1)
        wavelet_stage<WPC, MT>(P, L.xs);                // (everybody's share of the abscissae)
    // (one wave: wave 1 has the noise terms, wavelet_one_helper; four: waves 1..3 their share
    // of every likelihood)
    if (wave != 0) {
        wavelet_one_followers<MT, WPC>(P, L);
        return;
    }
    if (lane == 0)
        L.flag[3] = 0;
    if constexpr (WPC == 1)
        wavelet_stage<1, MT>(P, L.xs);
    double *np = L.nextp;
    if (lane < P.nparams)
        np[lane] = P.params[lane];
    if (lane + 64 < P.nparams)
        np[lane + 64] = P.params[lane + 64];
    block_sync<1>();
    double *Zr = own_global(P.Z), *log_post = own_global(P.log_post);
    const double *snap0 = own_global(P.snap0);
    int32_t *zchain = own_global(P.zchain);
    const int nchains = P.nchains, nf = P.nfree, zs = P.zs;
    const long long M0 = P.M0;
    const uint64_t seed = P.seed;
    const double gamma = P.gamma, fepsilon = P.fepsilon;
    const bool mine = lane < nf;
    const int kk = mine ? lane : 0;
    const long long R0 = P.ctl[0];
    double xk;
    {
        const double *snap_first = (R0 == 0) ? snap0 : Zr + (size_t)(M0 + (R0 - 1) * nchains) * zs;
        xk = mine ? snap_first[(size_t)c * zs + lane] : 0.0;
    }
    double cur_c = P.chisq[b], best = P.best_chisq[b];
    int nacc = 0;
    unsigned long long oob = 0;
    const int jfree = mine ? P.ifree[lane] : 0;
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    const bool need_normal = (SMP == MC3HIP_MRW) || (fepsilon != 0.0);
    const double pstep_k = (mine && need_normal) ? P.fstep[lane] : 0.0;
    const long long abort_at = (P.test_abort >= 0 && b == 0) ? (long long)P.test_abort : -1;
    auto normal_draw = [&](uint64_t it) {              // chain.py:185, as run_chain's
        double nk = 0.0;
        if (mine) {
            const Draw2 d = draw2(seed, it, (uint32_t)c, (uint32_t)(3 + (lane >> 1)));
            const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(d.q0)));
            double sn, cs;
            mc3_det_sincos2pi(u01(d.q1), &sn, &cs);
            nk = (rad * ((lane & 1) ? sn : cs)) * pstep_k;
        }
        return nk;
    };
    const double *ra[2] = {nullptr, nullptr};
    double rv[2] = {0.0, 0.0};
    auto request_rows = [&](uint64_t q0, long long R) {
        const long long Rs = R - 1;
        const double *snap = (Rs < 0) ? snap0 : Zr + (size_t)(M0 + Rs * nchains) * zs;
        int r1, r2;
        demc_partners(q0, nchains, c, r1, r2);
        ra[0] = snap + (size_t)r1 * zs + kk;
        ra[1] = snap + (size_t)r2 * zs + kk;
        rv[0] = load_published<false>(ra[0]);
        rv[1] = load_published<false>(ra[1]);
    };
    Draw2 nxt = draw2(seed, (uint64_t)R0, (uint32_t)c, 0u);
    if constexpr (SMP == MC3HIP_DEMC)
        request_rows(nxt.q0, R0);
    double normal_ahead = need_normal ? normal_draw((uint64_t)R0) : 0.0;
    for (int n = 0; n < nref; ++n) {
        const long long R = R0 + n;
        if (__builtin_expect(R == abort_at, 0)) {                      // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        const double u = u01(nxt.q1);                                  // chain.py:259
        const double normal_k = normal_ahead;
        nxt = draw2(seed, (uint64_t)(R + 1), (uint32_t)c, 0u);
        double lo, hi;
        accept_window(u, lo, hi);
        if (need_normal)
            normal_ahead = normal_draw((uint64_t)(R + 1));
        double jump_k;
        if constexpr (SMP == MC3HIP_MRW) {                             // chain.py:219-220
            jump_k = normal_k;
        } else {                                                       // chain.py:221-232
            if (!wait_rows<2, false>(ra, rv, ctl_rw))
                break;
            jump_k = gamma * (rv[0] - rv[1]) + fepsilon * normal_k;
        }
        const double npk = xk + jump_k;                                // chain.py:235-243
        const bool outk = mine && (npk < pmin_k || npk > pmax_k);
        bool accepted = false;
        if (!__any(outk)) {
            if (mine)
                np[jfree] = npk;
            const double total = wavelet_one_likelihood<MT, WPC>(P, L);
            // (as WaveletHook::finish: the prior term of a problem without priors is 0.0)
            const double nextchisq = is_pos_inf(total) ? total : total + 0.0;
            accepted = metropolis_accept(0.5 * (cur_c - nextchisq), u, lo, hi);
            if (accepted) {                                            // chain.py:257-274
                xk = npk;
                cur_c = nextchisq;
                nacc += 1;
            }
        } else {
            oob += outk ? 1ull : 0ull;                                 // chain.py:242-243
        }
        const long long row = M0 + R * nchains + c;                    // chain.py:276-289
        if (mine)
            __hip_atomic_store(&Zr[(size_t)row * zs + lane], xk, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);              // publish (sc1)
        // the partners' rows of this refresh: asked for right behind the own row
        if constexpr (SMP == MC3HIP_DEMC) {
            if (n + 1 < nref)
                request_rows(nxt.q0, R + 1);
        }
        if (lane == 0) {
            log_post[row] = -0.5 * cur_c;
            zchain[row] = c;
        }
        if (__builtin_expect(accepted && cur_c < best, 0)) {
            best = cur_c;
            if (mine)
                P.bestx[(size_t)b * nf + lane] = xk;
            if (lane == 0)
                P.best_iter[b] = R;
        }
    }
    wavelet_one_leave<WPC>(L);
    if (mine && oob)
        atomicAdd(&P.outbounds[lane], oob);
    if (mine)
        P.X[(size_t)b * nf + lane] = xk;
    if (lane == 0) {
        P.chisq[b] = cur_c;
        P.best_chisq[b] = best;
        P.numaccept[b] += (long long)nacc;
    }
}


}  // namespace mc3hip
