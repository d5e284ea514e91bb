// wavelet.h -- wavelet likelihood: the blocked D4 pyramid of one chain.
#pragma once
#include "core.h"
#include "chisq.h"

namespace mc3hip {

// ---- wavelet likelihood: blocked D4 pyramid -------------------------------------------
// Wavelet pseudo chi-squared (Carter & Winn 2009 eqs. 32-34), src_c/_dwt.c:76-114 over
// the D4 pyramid of src_c/include/wavelet.h:27-36,115-117, ndata = 2^M <= 32 T.
//
// T = 64 WPC threads share a chain.  Thread t owns the CONTIGUOUS block of B = N/Ta points
// [B t, B t + B) (Ta = min(T, N/2) threads take part; B is one of 2..32 and a compile-time
// constant of the code that runs, wavelet_body<.., B>).  The data values sit in LDS for
// the whole launch and the abscissae are recomputed (wavelet_stage below), so an
// iteration reads no global memory but the proposal:
//   A  log2 B levels inside the block, in registers; a level's halo (the next block's
//      first two values, wrapping like wavelet.h:35-36) comes through LDS, one raw
//      barrier per level from the second on; one value per thread is left;
//   B  (more than 64 left) those values, Ta/64 per lane, go through wave 0 the same way
//      -- no barrier, a wave's LDS traffic is ordered;
//   C  the last <= 64 values, one per lane, level by level through LDS, still wave 0.
// Detail coefficients are final when produced: an owner (thread / lane) adds the squares
// of its details of one level, acc = fma(d, d, acc), and folds the level in with the weight
// of its scale, w = fma(acc, 1/sigma_W^2, w); stage sums are halving trees, total
// (A + B) + C (the CPU checker states the same order).
// (Before: strided levels ping-ponging through 96 KiB of LDS, x and data re-read from L2
// every iteration, 13 barriers, half of the instruction stream index arithmetic.)
constexpr int WB = 32;                        // points per thread at most

// What a thread needs of the chain-invariant inputs is its block of B abscissae and B data
// values (plus the next block's first two of each).  One of the two arrays lives in LDS as
// [j][t] (lane-contiguous: conflict-free), staged once per launch:
//   * the DATA, when the abscissae need no storage: they lie on numpy.linspace's grid
//     (P.wl_grid, checked bit for bit at creation) and are recomputed, or the model is a
//     design matrix; nothing chain-invariant is then read from global memory in the loop;
//   * otherwise the ABSCISSAE, and the data come from global memory in the same [j][t]
//     layout (P.data_t; L2-resident, coalesced) -- the slower, general path.
// The caller synchronises the workgroup before the first use.
template <int WPC, int MT>
__device__ __forceinline__ void wavelet_stage(const DevProblem &P, double *xs)
{
    constexpr int T = 64 * WPC;
    const int N = P.ndata;
    const int Ta = (N / 2 < T) ? (N / 2 > 0 ? N / 2 : 1) : T;
    const int B = N / Ta;
    if ((int)threadIdx.x >= Ta || N > WB * T)
        return;                                 // (large data sets read global memory)
    const double *src = (MT == MT_LINEAR || P.wl_grid) ? P.data : P.indep;
    for (int j = 0; j < B; ++j)
        xs[j * T + threadIdx.x] = src[B * threadIdx.x + j];
}

// one level inside a block: v[0..NV) (+ halo at v[NV], v[NV+1]) -> v[0..NV/2), details
// squared into the returned sum; every index static
template <int NV, int CAP>
__device__ __forceinline__ double block_level(double (&v)[CAP])
{
    static_assert(NV + 2 <= CAP, "halo slots");
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < NV / 2; ++j) {
        double s, d;
        d4_pair(v[2 * j], v[2 * j + 1], v[2 * j + 2], v[2 * j + 3], s, d);
        v[j] = s;                              // (inputs of later pairs start at 2j + 2 > j)
        acc = mc3_fma(d, d, acc);
    }
    return acc;
}

// stage B / stage A from the second level on: levels NV -> NV/2 -> ... -> 1 with the halo
// exchanged through `hb` (two buffers of `stride` double2, indexed by level parity);
// SYNC = workgroup barrier (stage A) or none (one wave, stage B).  accs[LEV], accs[LEV+1],
// ... receive the levels' sums of squared details (weighted later, in level order).
#ifdef MC3HIP_PIPE_DIAG
__device__ unsigned long long mc3_pipe_diag[64];   // [0..7] level 0, [8] likelihoods, [16..23] stage A up to its last barrier, [24..31] from the slot's start to level 0's, [32..39] arrival at the first halo barrier, [40..47] at the second
#define PIPE_DIAG_ARG , unsigned long long dg0 = 0
#define PIPE_DIAG_PASS , dg0
#else
#define PIPE_DIAG_ARG
#define PIPE_DIAG_PASS
#endif
template <int NV, int CAP, int WPC, bool SYNC, int LEV, int NACC>
__device__ __forceinline__ void block_levels(double (&v)[CAP], double (&accs)[NACC],
                                             double2 *hb, int stride, int me, int next,
                                             bool active PIPE_DIAG_ARG)
{
    if constexpr (NV >= 2) {
        static_assert(LEV < NACC, "one slot per level");
        if constexpr (SYNC) {
            double2 *h = hb + (LEV & 1) * stride;
            if (active)
                h[me] = make_double2(v[0], v[1]);
#ifdef MC3HIP_PIPE_DIAG
            if (dg0 != 0 && LEV <= 2) {
                // (everything this wave has issued is done -- THEN the clock)
#pragma unroll
                for (int q = 0; q < NV; ++q)
                    asm volatile("; done" : "+v"(v[q]));
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
                const unsigned long long now = 10ull * __builtin_amdgcn_s_memrealtime();
                if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)
                    atomicAdd(&mc3_pipe_diag[32 + 8 * (LEV - 1) + (threadIdx.x >> 6)], now - dg0);
            }
#endif
            block_sync<WPC>();
            const double2 hv = h[next];
            v[NV] = hv.x;
            v[NV + 1] = hv.y;
        } else {
            // one wave, every lane taking part, the next block is the next lane's: the halo
            // comes by a lane rotation instead of a round trip through LDS
            v[NV] = wave_rol1(v[0]);
            v[NV + 1] = wave_rol1(v[1]);
        }
        accs[LEV] = block_level<NV, CAP>(v);
        block_levels<NV / 2, CAP, WPC, SYNC, LEV + 1, NACC>(v, accs, hb, stride, me, next, active
                                                            PIPE_DIAG_PASS);
    }
}

constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n / 2); }

// the levels' sums of squared details folded in with the weights of their scales
// (1/sigma_W^2 of scale m0, m0 - 1, ...), in level order
template <int NLEV, int NACC>
__device__ __forceinline__ double weigh_levels(const double (&accs)[NACC], const double *s_tab,
                                               int m0, double w = 0.0)
{
#pragma unroll
    for (int l = 0; l < NLEV; ++l)
        w = mc3_fma(accs[l], s_tab[m0 - l], w);
    return w;
}

// Level 0 of stage A: streams the residuals data - model (src_c/_dwt.c:89) of the block,
// two new points per output pair, so that only the B/2 smooth values are ever held.
// GRID: x recomputed on np.linspace's grid, data from LDS; otherwise x from LDS, data from
// the register-resident tile.  Returns the level's sum of squared details.
// VALUES: the block's inputs are values already (the output of the pre-levels of a large
// data set, in global memory as a plain array), not residuals to be formed.
// Only the first P.wl_nreal points are data (normally all; otherwise the rest pads the series to a
// power of two, the reference's zero padding src_c/_dwt.c:82-91 made well defined:
// mc3hip_problem.wl_nreal): the residual of a padding point is exactly zero, whatever the model
// says there.  A padded series is never on a grid, so only the general flavour (GRID = false)
// looks -- a compare and a select per point there, nothing on the grid flavour C4 runs.
template <int WPC, int MT, int B, bool GRID, int CAP, bool VALUES = false>
__device__ __forceinline__ double wavelet_level0(const DevProblem &P, const Model<MT> &model,
                                                 const double *xs, int me, int next, int Ta,
                                                 int &sawinf, double (&v)[CAP],
                                                 const double *values = nullptr)
{
    constexpr int T = 64 * WPC;
    const int N = VALUES ? B * Ta : P.ndata;
    const int base = B * me;
    double fbase = (double)base, fnext = (double)(B * next);
    int ome = me, onext = next;
    // All of this is invariant across the iterations of a persistent launch, and the
    // optimiser would hoist it out of the chain loop -- 2 B values per thread parked in
    // registers (and spilled) for good.  The empty asm makes the inputs opaque per call
    // (offsets, not pointers: a laundered pointer loses its address space and the LDS
    // reads would become flat loads).
    asm volatile("; per-iteration inputs" : "+v"(fbase), "+v"(fnext), "+v"(ome), "+v"(onext));
    const double *xme = xs + ome, *xnext = xs + onext;
    const double *dme = P.data_t + ome, *dnext = P.data_t + onext;
    // what a point needs from LDS / global memory is requested PF points ahead of its use:
    // far enough to cover the latency behind the arithmetic of the points in between,
    // near enough to keep only a few values in flight (registers)
    constexpr int PF = 8;
    double in0[B + 2], in1[(MT != MT_LINEAR && !GRID) ? B + 2 : 1];
    auto fetch = [&](int j) {                           // static j
        if constexpr (VALUES) {
            in0[j] = (j < B) ? values[B * ome + j] : values[B * onext + (j - B)];
        } else {
            in0[j] = (j < B) ? xme[j * T] : xnext[(j - B) * T];
            if constexpr (MT != MT_LINEAR && !GRID)
                in1[j] = (j < B) ? dme[j * T] : dnext[(j - B) * T];
        }
    };
#pragma unroll
    for (int j = 0; j < PF + 2 && j < B + 2; ++j)
        fetch(j);
    (void)in1;
    // Design-matrix models: the model values of the block (and of the two halo points), column
    // by column -- Model::at's order per point, k ascending -- so that a column's values are
    // consumed as they arrive instead of every point running its own loop over the columns
    // with all their loads in flight (121 registers spilled in the nine-wave kernel).
    double mv[(MT == MT_LINEAR && !VALUES) ? B + 2 : 1];
    if constexpr (MT == MT_LINEAR && !VALUES) {
        const int pnext = (base + B) & (N - 1);
        const double *col = model.indep;
        {
            const double c0 = model.coef[0];
#pragma unroll
            for (int j = 0; j < B + 2; ++j)
                mv[j] = col[j < B ? base + j : pnext + (j - B)] * c0;
        }
        for (int k = 1; k < model.nmodel; ++k) {
            col += model.ndata;
            const double ck = model.coef[k];
#pragma unroll
            for (int j = 0; j < B + 2; ++j)
                mv[j] = mc3_fma(col[j < B ? base + j : pnext + (j - B)], ck, mv[j]);
        }
    }
    (void)mv;
    // the model at point j of the block (B, B+1: the next block's first two); j static in the
    // pyramid below, a loop index in the rare rescan
    auto model_value = [&](int j, double xj) {
        if constexpr (MT == MT_LINEAR) {
            return model.at(j < B ? base + j : (base + j) & (N - 1));
        } else if constexpr (GRID) {
            // numpy.linspace: y = arange(num) * step + start, y[-1] = stop
            const double fi = (j < B) ? fbase + (double)j : fnext + (double)(j - B);
            double x = fi * P.wl_step + P.wl_x0;
            if (j == B - 1)
                x = (me == Ta - 1) ? P.wl_xlast : x;
            if (j >= B && N <= 2)
                x = (j - B == N - 1) ? P.wl_xlast : x;
            return model.at_x(x);
        } else {
            return model.at_x(xj);
        }
    };
    // A +inf model value makes the likelihood +inf (chain.py:322-323).  Not looked for point
    // by point: the residual of such a point is -inf (or NaN), every D4 coefficient is
    // non-zero, so the detail of its pair and with it this level's sum of squares is not
    // finite -- one test per block below, and only then are the block's model values (and the
    // two halo points', which enter its last pair) looked at one by one.
    // points of the block, and of the next block's first two, that are data
    constexpr bool PAD = !GRID && !VALUES;
    const int nreal_me = PAD ? P.wl_nreal - base : B;
    const int nreal_next = PAD ? P.wl_nreal - B * next : 2;
    (void)nreal_me; (void)nreal_next;
    auto resid_of = [&](int j) {                        // j: static index into the block
        if constexpr (VALUES) {
            return in0[j];
        } else if constexpr (MT == MT_LINEAR) {
            return in0[j] - mv[j];
        } else if constexpr (GRID) {
            return in0[j] - model_value(j, 0.0);
        } else {
            return in1[j] - model_value(j, in0[j]);
        }
    };
    auto resid = [&](int j) {
        const double r = resid_of(j);
        if constexpr (PAD)
            return ((j < B) ? j < nreal_me : j - B < nreal_next) ? r : 0.0;
        else
            return r;
    };
    auto rescan = [&]() {
        int seen = 0;
        if constexpr (!VALUES) {
#pragma unroll 1
            for (int j = 0; j < B + 2; ++j) {
                double xj = 0.0;
                if constexpr (MT != MT_LINEAR && !GRID)
                    xj = (j < B) ? xme[j * T] : xnext[(j - B) * T];
                seen |= is_pos_inf(model_value(j, xj)) ? 1 : 0;
            }
        }
        return seen;
    };
    double acc0 = 0.0;
    if (N >= 4) {
        double r0 = resid(0), r1 = resid(1);
#pragma unroll
        for (int j = 0; j < B / 2; ++j) {
            if (2 * j + 2 + PF < B + 2) {
                fetch(2 * j + 2 + PF);
                fetch(2 * j + 3 + PF);
            }
            const double r2 = resid(2 * j + 2), r3 = resid(2 * j + 3);
            double s, d;
            d4_pair(r0, r1, r2, r3, s, d);
            v[j] = s;
            acc0 = mc3_fma(d, d, acc0);
            r0 = r2;
            r1 = r3;
        }
        if (__builtin_expect(!(__builtin_fabs(acc0) < __builtin_inf()), 0))
            sawinf |= rescan();
    } else {
        v[0] = resid(0);                                // N = 2: no level at all
        v[1] = resid(1);
        if (!(__builtin_fabs(v[0]) < __builtin_inf()) || !(__builtin_fabs(v[1]) < __builtin_inf()))
            sawinf |= rescan();
    }
    return acc0;
}

// The per-scale noise terms depend only on the three noise parameters: lane m computes
// sigma_W^2(m), its reciprocal (table in LDS) and its share of the log-determinant, lane
// 0 the smooth coefficients' sigma_S^2 -- one exp2 + log per lane instead of M of each on
// every thread -- and the log-determinant is summed in scale order.  Results in LDS:
// s_tab[1..M-1] = 1/sigma_W^2 (M <= 22), s_tab[24] = sigma_S^2, s_tab[25] = the
// log-determinant.
__device__ __forceinline__ void wavelet_noise_terms(const double *s_nextp, int nparams, int M,
                                                    double *s_tab)
{
    const int lane = threadIdx.x & 63;
    const double gam = s_nextp[nparams - 3];
    const double sr = s_nextp[nparams - 2];
    const double sw = s_nextp[nparams - 1];
    const double sr2 = sr * sr, sw2 = sw * sw;
    // (one pass for the smooth scale -- lane 0 -- and the detail scales: the same operations
    // on each lane's own operands as two branches would do, at half the instructions)
    const bool smooth = lane == 0;
    const double e = mc3_det_exp2(smooth ? -gam : -gam * (double)lane);
    const double se = sr2 * e;
    const double s2 = (smooth ? se * 0.72134752 : se) + sw2;     // sigma_S^2 / sigma_W^2
    const double lg = mc3_det_log(6.283185307179586 * s2);
    double term = 0.0;
    if (lane >= 1 && lane < M) {
        s_tab[lane] = 1.0 / s2;
        term = (double)(1 << (lane & 31)) * lg;
    } else if (smooth) {
        s_tab[24] = s2;
        term = 2.0 * lg;
    }
    double Lg = read_lane(term, 0);
    for (int m = 1; m < M; ++m)
        Lg = Lg + read_lane(term, m);             // scales added in order
    if (lane == 0)
        s_tab[25] = Lg;
}

// `values` (large data sets): the blocks are cut from that array of B T values -- the
// output of `npre` pre-levels whose per-thread sums of squared details arrive in accP --
// instead of from the residuals.
// CW (chain-wave kernels, kernels_wavelet.h): the workgroup has one more wave, which runs the
// chain and has left the noise terms in s_tab BEFORE the first halo barrier of stage A (it
// takes part in every workgroup barrier here, doing nothing in between): a thread folds its
// stage-A sums with the weights right away, and the two barriers that waited for the noise
// terms and for those sums are gone.
// PIPE (kernels_wavelet.h wavelet_pipe_kernel): stage A only -- everything behind barrier (1) is
// run by a wave of its own (PipeTail below) while these waves start on the next likelihood.
// NX (one wave): the noise terms are a HELPER wave's work -- it is computing them while this
// wave transforms, and a barrier of the two stands where they are first used (stage A's sums,
// behind the whole transform: ~1.1 us of dependent software exp2 / log / the log-determinant's
// serial additions out of the chain's way).
template <int WPC, int MT, int B, bool VALUES = false, bool CW = false, bool PIPE = false,
          bool NX = false>
__device__ __forceinline__ double wavelet_body(const DevProblem &P, const double *s_nextp,
                                               double *s_part, int *s_flag, double *s_tab,
                                               double *s_halo, const double *xs,
                                               const double *values = nullptr, int npre = 0,
                                               const double *accP = nullptr, int sawinf0 = 0,
                                               const Model<MT> *ready_model = nullptr
#ifdef MC3HIP_PIPE_DIAG
                                               , unsigned long long dg_slot0 = 0
#endif
                                               )
{
    constexpr int T = 64 * WPC;
    constexpr int NLA = ilog2c(B);                      // levels of stage A (N >= 4)
    constexpr int NLB = ilog2c(WPC);                    // levels of stage B
    const int tid = threadIdx.x, lane = tid & 63;
    // (made wave-uniform by hand so that the per-wave roles below are scalar branches)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the wave that runs stages B and C alone.  Chain-wave kernels: not wave 0 -- the ninth
    // wave lands on wave 0's SIMD (waves are dealt round robin, tools/probes/wave_placement.hip),
    // and what it prepares for the next iteration while the tail runs would take issue slots
    // from the one wave everybody is waiting for
    constexpr int TW = (CW && WPC > 1) ? 1 : 0;
    const int M = ilog2(P.ndata);
    const int N = VALUES ? B * T : P.ndata;             // length of the array stage A starts from
    const int Ta = N / B;                               // threads that own a block
    const bool active = tid < Ta;
    const int me = active ? tid : 0;
    const int next = (me + 1 == Ta) ? 0 : me + 1;
    Model<MT> model;
    if (CW && ready_model != nullptr)
        model = *ready_model;                           // (read together with the order to go)
    else
        model.init(P, s_nextp);
    if (WPC > 1 && tid == 0)
        s_flag[1] = 0;                                  // "a model value was +inf"
    // One wave: the noise terms first.  Several waves: the LAST wave computes them after its
    // share of stage A, while wave 0 is busy with stages B and C -- two long dependent chains
    // (software exp2 / log; cross-lane levels) side by side instead of one after the other.
    static_assert(!NX || WPC == 1, "");
    if constexpr (WPC == 1 && !NX)
        wavelet_noise_terms(s_nextp, P.nparams, M, s_tab);
#ifdef MC3HIP_PIPE_DIAG
    const unsigned long long dg_in = (10ull * __builtin_amdgcn_s_memrealtime());
#endif
    MC3_LSTAMP(8);
    // ---- stage A: log2 B levels in registers ------------------------------------------
    constexpr int HB = (B / 2 > 2) ? B / 2 : 2;
    double v[HB + 2];
    double accA[NLA > 0 ? NLA : 1];
    int sawinf = sawinf0;
    int nn = N;
    if constexpr (VALUES)
        accA[0] = wavelet_level0<WPC, MT, B, false, HB + 2, true>(P, model, xs, me, next, Ta,
                                                                  sawinf, v, values);
    else if (MT != MT_LINEAR && P.wl_grid)                  // (never a padded series)
        accA[0] = wavelet_level0<WPC, MT, B, true, HB + 2>(P, model, xs, me, next, Ta, sawinf, v);
    else
        accA[0] = wavelet_level0<WPC, MT, B, false, HB + 2>(P, model, xs, me, next, Ta, sawinf, v);
#ifdef MC3HIP_PIPE_DIAG
    if (PIPE && blockIdx.x == 0 && (tid & 63) == 0) {
        const unsigned long long now = (10ull * __builtin_amdgcn_s_memrealtime());
        atomicAdd(&mc3_pipe_diag[tid >> 6], now - dg_in);
        atomicAdd(&mc3_pipe_diag[24 + (tid >> 6)], dg_in - dg_slot0);
        if (tid == 0)
            atomicAdd(&mc3_pipe_diag[8], 1ull);
    }
#endif
    MC3_LSTAMP(14);
    double2 *halo = reinterpret_cast<double2 *>(s_halo);  // [2][T]
    if (N >= 4) {
#ifdef MC3HIP_PIPE_DIAG
        block_levels<B / 2, HB + 2, WPC, true, 1, (NLA > 0 ? NLA : 1)>(v, accA, halo, T, me, next,
                                                                        active, PIPE ? dg_slot0 : 0);
#else
        block_levels<B / 2, HB + 2, WPC, true, 1, (NLA > 0 ? NLA : 1)>(v, accA, halo, T, me, next,
                                                                        active);
#endif
        nn = Ta;                                        // one value per thread
    }
    MC3_LSTAMP(9);
    const int mA = M - 1 - npre;                        // scale of stage A's first level
    // (large data sets) the pre-levels' sums come first, in level order
    auto stage_a_sum = [&]() {
        double w0 = 0.0;
#pragma unroll
        for (int l = 0; l < 8; ++l)
            if (VALUES && l < npre)
                w0 = mc3_fma(accP[l], s_tab[M - 1 - l], w0);
        return weigh_levels<NLA>(accA, s_tab, mA, w0);
    };
    double sumA, u = v[0];
    double pA = 0.0;                                    // CW: this lane's share of stage A's sum
    int mscale = (N >= 4) ? mA - NLA : mA;              // scale of the next level's details
    double accB[NLB > 0 ? NLB : 1];
    if constexpr (WPC > 1) {
        // (a buffer of its own: the last halo exchange may still be being read by a wave
        // that is behind)
        double *ubuf = s_halo + 4 * T + 64;             // [T]
        ubuf[tid] = v[0];
        if (sawinf)
            s_flag[1] = 1;
        if constexpr (CW)
            s_part[tid] = (active && N >= 4) ? stage_a_sum() : 0.0;
#ifdef MC3HIP_PIPE_DIAG
        if (PIPE && blockIdx.x == 0 && (tid & 63) == 0)
            atomicAdd(&mc3_pipe_diag[16 + (tid >> 6)], (10ull * __builtin_amdgcn_s_memrealtime()) - dg_in);
#endif
        block_sync<WPC>();                              // (1) the T values are in LDS
        if constexpr (PIPE)
            return 0.0;
        if (!CW && wave == WPC - 1)
            wavelet_noise_terms(s_nextp, P.nparams, M, s_tab);
        if (CW && wave == TW) {
            // (chain-wave kernels: everybody's stage-A sums are in LDS already -- asked for now,
            // added up while stage B runs)
            double p[WPC];
#pragma unroll
            for (int q = 0; q < WPC; ++q)
                p[q] = s_part[lane + 64 * q];
#pragma unroll
            for (int s2 = WPC / 2; s2 >= 1; s2 >>= 1) {
#pragma unroll
                for (int q = 0; q < s2; ++q)
                    p[q] = p[q] + p[q + s2];
            }
            pA = p[0];
        }
        if (wave == TW) {
            // ---- stage B: T values, WPC per lane, levels inside the lane -----------------
            double vb[WPC + 2];
#pragma unroll
            for (int j = 0; j < WPC; ++j)
                vb[j] = ubuf[WPC * lane + j];
            const int hbase = (WPC * (lane + 1)) & (T - 1);
            vb[WPC] = ubuf[hbase];
            vb[WPC + 1] = ubuf[hbase + 1];
            accB[0] = block_level<WPC, WPC + 2>(vb);
            block_levels<WPC / 2, WPC + 2, 1, false, 1, (NLB > 0 ? NLB : 1)>(
                vb, accB, halo + T, 64, lane, (lane + 1) & 63, true);
            u = vb[0];
        }
        mscale -= NLB;
        nn = 64;
    }
    MC3_LSTAMP(10);
    // ---- stage C: nn <= 64 values, one per lane (wave 0) ------------------------------------
    double accC[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int mC = mscale;
    int nC = 0;
    if (wave == TW) {
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            if (nn >= 4) {
                const int nh = nn >> 1;
                // lane l < nh takes values 2l .. 2l + 3 (wrapping) straight from the lanes that
                // hold them (the LDS crossbar: no store, no wait for it, no read-back)
                const int i2 = (lane < nh) ? 2 * lane : 0;
                const int i3 = (i2 + 2) & (nn - 1);
                const double c0 = lane_gather(u, i2), c1 = lane_gather(u, i2 + 1);
                const double c2 = lane_gather(u, i3), c3 = lane_gather(u, i3 + 1);
                double s, d;
                d4_pair(c0, c1, c2, c3, s, d);
                accC[l] = (lane < nh) ? d * d : 0.0;
                u = s;
                nn = nh;
                nC = l + 1;
            }
        }
    }
    MC3_LSTAMP(12);
    double Tsum;
    if constexpr (WPC > 1) {
        if constexpr (!CW) {
            block_sync<WPC>();                          // (2) the noise terms are in LDS
            s_part[tid] = (active && N >= 4) ? stage_a_sum() : 0.0;
            block_sync<WPC>();                          // (3) every thread's stage-A sum
        }
        if (wave != TW)
            return 0.0;                                 // the value is only used by that wave
        if constexpr (CW) {
            // the three stage sums in one interleaved pass; total (A + B) + C as ever
            double sA = pA, sB = weigh_levels<NLB>(accB, s_tab, mA - NLA), sC = 0.0;
#pragma unroll
            for (int l = 0; l < 6; ++l)
                if (l < nC)
                    sC = mc3_fma(accC[l], s_tab[mC - l], sC);
            wave_sum3(sA, sB, sC);
            sawinf = s_flag[1];
            const double a0c = read_lane(u, 0), a1c = read_lane(u, 1);
            MC3_LSTAMP(13);
            if (sawinf)
                return mc3_from_bits(0x7ff0000000000000ull);
            const double sS2c = s_tab[24];
            const double Sc = (a0c * a0c) / sS2c + (a1c * a1c) / sS2c;
            return (Sc + s_tab[25]) + ((sA + sB) + sC);
        }
        double p[WPC];
#pragma unroll
        for (int q = 0; q < WPC; ++q)
            p[q] = s_part[lane + 64 * q];
#pragma unroll
        for (int s2 = WPC / 2; s2 >= 1; s2 >>= 1) {
#pragma unroll
            for (int q = 0; q < s2; ++q)
                p[q] = p[q] + p[q + s2];
        }
        sumA = wave_sum(p[0]);
        sawinf = s_flag[1];
        sumA = sumA + wave_sum(weigh_levels<NLB>(accB, s_tab, mA - NLA));      // (A + B)
    } else {
        if constexpr (NX)
            block_sync<2>();                            // the helper wave's noise terms are in LDS
        sumA = wave_sum((active && N >= 4) ? stage_a_sum() : 0.0);
        sawinf = __any(sawinf) ? 1 : 0;
    }
    {
        double wc = 0.0;
#pragma unroll
        for (int l = 0; l < 6; ++l)
            if (l < nC)
                wc = mc3_fma(accC[l], s_tab[mC - l], wc);
        Tsum = sumA + wave_sum(wc);
    }
    // the two smooth coefficients left (N = 2: the two residuals, both in thread 0)
    const double a0 = read_lane(u, 0), a1 = (N == 2) ? read_lane(v[1], 0) : read_lane(u, 1);
    MC3_LSTAMP(13);
    if (sawinf)
        return mc3_from_bits(0x7ff0000000000000ull);
    const double sS2 = s_tab[24];
    const double S = (a0 * a0) / sS2 + (a1 * a1) / sS2;
    return (S + s_tab[25]) + Tsum;
}

// Everything of wavelet_body<WPC, MT, B, false, CW> behind barrier (1), for ONE wave that is not
// one of the WPC: load() between (1) and the next likelihood's first barrier (the stage-A sums,
// the T values and the flag are overwritten by the next stage A), run() whenever -- the same
// arithmetic in the same order as the tail above, so the value is bit-identical.
template <int WPC>
struct PipeTail {
    double vb[WPC + 2];
    double pA;
    int sawinf;

    __device__ __forceinline__ void load(const double *s_part, const double *s_halo,
                                         const int *s_flag)
    {
        constexpr int T = 64 * WPC;
        const int lane = threadIdx.x & 63;
        const double *ubuf = s_halo + 4 * T + 64;
        double p[WPC];
#pragma unroll
        for (int q = 0; q < WPC; ++q)
            p[q] = s_part[lane + 64 * q];
#pragma unroll
        for (int j = 0; j < WPC; ++j)
            vb[j] = ubuf[WPC * lane + j];
        const int hbase = (WPC * (lane + 1)) & (T - 1);
        vb[WPC] = ubuf[hbase];
        vb[WPC + 1] = ubuf[hbase + 1];
        sawinf = s_flag[1];
#pragma unroll
        for (int s2 = WPC / 2; s2 >= 1; s2 >>= 1) {
#pragma unroll
            for (int q = 0; q < s2; ++q)
                p[q] = p[q] + p[q + s2];
        }
        pA = p[0];
    }

    template <int B>
    __device__ __forceinline__ double run(const double *s_tab, const int M)
    {
        constexpr int NLA = ilog2c(B), NLB = ilog2c(WPC);
        const int lane = threadIdx.x & 63;
        const int mA = M - 1;
        double accB[NLB];
        accB[0] = block_level<WPC, WPC + 2>(vb);
        block_levels<WPC / 2, WPC + 2, 1, false, 1, NLB>(vb, accB, nullptr, 64, lane,
                                                         (lane + 1) & 63, true);
        double u = vb[0];
        const int mC = mA - NLA - NLB;
        int nn = 64, nC = 0;
        double accC[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            if (nn >= 4) {
                const int nh = nn >> 1;
                const int i2 = (lane < nh) ? 2 * lane : 0;
                const int i3 = (i2 + 2) & (nn - 1);
                const double c0 = lane_gather(u, i2), c1 = lane_gather(u, i2 + 1);
                const double c2 = lane_gather(u, i3), c3 = lane_gather(u, i3 + 1);
                double s, d;
                d4_pair(c0, c1, c2, c3, s, d);
                accC[l] = (lane < nh) ? d * d : 0.0;
                u = s;
                nn = nh;
                nC = l + 1;
            }
        }
        double sA = pA, sB = weigh_levels<NLB>(accB, s_tab, mA - NLA), sC = 0.0;
#pragma unroll
        for (int l = 0; l < 6; ++l)
            if (l < nC)
                sC = mc3_fma(accC[l], s_tab[mC - l], sC);
        wave_sum3(sA, sB, sC);
        const double a0c = read_lane(u, 0), a1c = read_lane(u, 1);
        if (sawinf)
            return mc3_from_bits(0x7ff0000000000000ull);
        const double sS2c = s_tab[24];
        const double Sc = (a0c * a0c) / sS2c + (a1c * a1c) / sS2c;
        return (Sc + s_tab[25]) + ((sA + sB) + sC);
    }
};

// More than 32 points per thread (ndata > 2^14): "pre-levels" bring the pyramid down to
// 2^14 values first.  Output i of a pre-level belongs to thread i mod T (strided, coalesced);
// level 0 forms the residuals on the fly from global memory, the smooth values ping-pong
// between two global buffers of this workgroup (P.wl_scratch); each thread keeps its sum of
// squared details per pre-level, weighted with the others at the end.  Then the blocked
// pyramid above runs on the 2^14 values.  Slower per point than the resident path (every
// level goes through global memory), but any 2^M up to 2^22 fits.
template <int MT>
__device__ double wavelet_large(const DevProblem &P, const double *s_nextp, double *s_part,
                                int *s_flag, double *s_tab, double *s_halo, const double *xs)
{
    constexpr int T = 512;
    const int tid = threadIdx.x;
    const int N = P.ndata;
    Model<MT> model;
    model.init(P, s_nextp);
    double *bufA = P.wl_scratch + (size_t)blockIdx.x * (3 * (size_t)N / 4), *bufB = bufA + N / 2;
    double accP[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int sawinf = 0, npre = 0, nn = N;
    {
        const int nh = nn >> 1;
        double acc = 0.0;
        for (int i = tid; i < nh; i += T) {
            const int j0 = 2 * i, j2 = (2 * i + 2) & (N - 1);
            double m0, m1, m2, m3;
            if constexpr (MT == MT_LINEAR) {
                m0 = model.at(j0);
                m1 = model.at(j0 + 1);
                m2 = model.at(j2);
                m3 = model.at(j2 + 1);
            } else {
                const double2 xa = *reinterpret_cast<const double2 *>(P.indep + j0);
                const double2 xb = *reinterpret_cast<const double2 *>(P.indep + j2);
                m0 = model.at_x(xa.x);
                m1 = model.at_x(xa.y);
                m2 = model.at_x(xb.x);
                m3 = model.at_x(xb.y);
            }
            sawinf |= is_pos_inf(m0) | is_pos_inf(m1) | is_pos_inf(m2) | is_pos_inf(m3);
            const double2 da = *reinterpret_cast<const double2 *>(P.data + j0);
            const double2 db = *reinterpret_cast<const double2 *>(P.data + j2);
            double s2, d;
            // (a padded series, P.wl_nreal < ndata: the padding points' residuals are zero)
            const int nr = P.wl_nreal;
            d4_pair(j0 < nr ? da.x - m0 : 0.0, j0 + 1 < nr ? da.y - m1 : 0.0,
                    j2 < nr ? db.x - m2 : 0.0, j2 + 1 < nr ? db.y - m3 : 0.0, s2, d);
            bufA[i] = s2;
            acc = mc3_fma(d, d, acc);
        }
        accP[0] = acc;
        npre = 1;
        nn = nh;
    }
    __syncthreads();                 // (also makes the global stores visible to the workgroup)
    double *cur = bufA, *dst = bufB;
#pragma unroll
    for (int l = 1; l < 8; ++l) {
        if (nn > WB * T) {
            const int nh = nn >> 1;
            double acc = 0.0;
            for (int i = tid; i < nh; i += T) {
                const double2 ca = *reinterpret_cast<const double2 *>(cur + 2 * i);
                const double2 cc = *reinterpret_cast<const double2 *>(cur + ((2 * i + 2) & (nn - 1)));
                double s2, d;
                d4_pair(ca.x, ca.y, cc.x, cc.y, s2, d);
                dst[i] = s2;
                acc = mc3_fma(d, d, acc);
            }
            accP[l] = acc;
            npre = l + 1;
            nn = nh;
            double *t2 = cur;
            cur = dst;
            dst = t2;
            __syncthreads();
        }
    }
    return wavelet_body<8, MT, WB, true>(P, s_nextp, s_part, s_flag, s_tab, s_halo, xs, cur, npre,
                                         accP, sawinf);
}

// The geometry is a function of ndata alone (mc3hip.hip choose_wpc): one wave up to 2048
// points (B = ndata/64, at least 2), four waves at 4096 (B = 16), eight above (B = 16, 32).
// LARGE (a kernel instance of its own, so that the resident path keeps its register
// allocation): ndata > 2^14, eight waves, pre-levels through global memory.
// (declared, with LARGE = false by default, in chisq.h)
template <int WPC, int MT, bool LARGE, bool NX>
__device__ double block_wavelet(const DevProblem &P, const double *s_nextp, double *s_part,
                                int *s_flag, double *s_tab, double *s_halo, const double *xs)
{
    static_assert(!NX || (WPC == 1 && !LARGE), "");
    constexpr int T = 64 * WPC;
    if constexpr (LARGE) {
        static_assert(WPC == 8, "the large-data path has eight waves");
        return wavelet_large<MT>(P, s_nextp, s_part, s_flag, s_tab, s_halo, xs);
    }
    const int B = (P.ndata / 2 < T) ? 2 : P.ndata / T;
#define MC3_WL_BODY(BB)                                                                       \
    return wavelet_body<WPC, MT, BB, false, false, false, NX>(P, s_nextp, s_part, s_flag, s_tab, \
                                                              s_halo, xs)
    if constexpr (WPC == 8) {
        if (B == 32)
            MC3_WL_BODY(32);
        MC3_WL_BODY(16);
    } else if constexpr (WPC == 4) {
        MC3_WL_BODY(16);
    } else {
        switch (B) {
        case 32: MC3_WL_BODY(32);
        case 16: MC3_WL_BODY(16);
        case 8: MC3_WL_BODY(8);
        case 4: MC3_WL_BODY(4);
        default: MC3_WL_BODY(2);
        }
    }
#undef MC3_WL_BODY
}

}  // namespace mc3hip
