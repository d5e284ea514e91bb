// tiled.h -- tiled plan (large ndata): per-tile data sums for every chain, one launch.
#pragma once
#include "core.h"
#include "run_chain.h"

namespace mc3hip {

// ---- tiled plan: data sums for every chain, one launch ------------------------
// For large ndata the fused kernel would make every chain stream the whole data
// set (and the design matrix) through its own workgroup.  Here a workgroup keeps
// ONE tile of (A or x, data, w) in registers -- 1024 threads, thread l owns virtual
// lane l with its TP points (tile = 1024*TP: 4096 for <= 4 coefficients, 2048 above,
// so that tile + 16 partial sums fit the 128-register budget of a 1024-thread group) -- and loops over a block of chains, so
// the tile is read once per launch instead of once per chain.  Chains are taken
// 16 at a time: every thread produces 16 partial sums, one LDS exchange, then
// wave g reduces chain g over the 1024 lanes in the canonical order (strides
// 512..64 across waves, 32..1 inside the wave).  Output: sc_part[chain][tile];
// the finish kernel adds the tiles' sums as the wave's halving tree (chisq.h finish_chisq).
// KX > 0: the number of coefficients is known at compile time (polynomials of 2..4
// coefficients, the same family the fused kernels specialise): the Horner chain is KX - 1
// fused multiply-adds instead of KMAX steps of "is this coefficient there?" selects, which
// made a quadratic cost five times its arithmetic (58 -> 14 us per launch on 512 chains x
// 1e5 points).
template <int KMAX, bool LINEAR, int TP, int KX = 0>
__global__ __launch_bounds__(1024) void tile_chisq_kernel(
    const DevProblem P, const int chain_block,
    // separate read-only kernel arguments so that the wave-uniform coefficient reads
    // become scalar loads (s_load) instead of 64-lane vector loads
    const double *__restrict__ nextp, const int32_t *__restrict__ out_of_bounds)
{
    extern __shared__ __attribute__((aligned(16))) double s_x[];   // [16][1024]
    const int l = threadIdx.x, lane = l & 63, wv = l >> 6;
    const int t = blockIdx.x;
    const int K = KX > 0 ? KX : P.nmodel;
    static_assert(KX == 0 || (!LINEAR && KX >= 2 && KX <= KMAX), "exact degree: polynomials");
    const size_t base = (size_t)t * (1024 * TP) + l;      // TP points per thread
    double a[TP][LINEAR ? KMAX : 1], d[TP], w[TP];
#pragma unroll
    for (int q = 0; q < TP; ++q) {
        const size_t i = base + 1024 * q;
        d[q] = P.data[i];
        w[q] = P.w[i];
        if constexpr (LINEAR) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                a[q][k] = (k < K) ? P.indep[(size_t)k * P.npad + i] : 0.0;
        } else {
            a[q][0] = P.indep[i];
        }
    }
    const int c0 = blockIdx.y * chain_block;
    const int c1 = (c0 + chain_block < P.nlocal) ? c0 + chain_block : P.nlocal;
    // The 16 coefficient vectors of a block of chains go through LDS: one cooperative copy
    // per block (16 x KMAX doubles, double-buffered so that the copy of the next block rides
    // on this block's barriers), read back as same-address broadcasts.  Scalar loads of each
    // chain's row straight from global memory stalled the wave 16 times per block
    // (s_load_dwordx16 + s_waitcnt in front of ~30 fused multiply-adds: 30 % of the cycles).
    double *s_cf = s_x + 16 * 1024;                       // [2][16][KMAX + 2] (+ flag)
    constexpr int CFS = KMAX + 2;
    auto stage_coefficients = [&](int cb, int buf) {
        if (l < 16 * CFS) {
            const int g = l / CFS, k = l - g * CFS;
            const int lc = cb + g;
            double v = 0.0;
            if (lc < c1) {
                if (k < KMAX)
                    v = nextp[(size_t)lc * SC_STRIDE + k];
                else if (k == KMAX)
                    v = (out_of_bounds[lc] & 1) ? 1.0 : 0.0;      // skip flag
            } else if (k == KMAX) {
                v = 1.0;
            }
            s_cf[(buf * 16 + g) * CFS + k] = v;
        }
    };
    // KX > 0 (KMAX = 4): the 16 x 4 coefficients of a block are exactly one value per lane;
    // every wave loads them itself (lane 4 g + k: coefficient k of chain cb + g; the next
    // block's while this one is computed) and hands them to the scalar unit with v_readlane --
    // no LDS staging, no 80 broadcast reads per block through the LDS pipe all 16 waves share.
    double cf_lanes = 0.0, cf_next = 0.0;
    int skip_lanes = 1, skip_next = 1;
    auto request_coefficients = [&](int cb, double &cfv, int &skipv) {
        const int g = lane >> 2, lc = cb + g;
        cfv = (lc < c1) ? nextp[(size_t)lc * SC_STRIDE + (lane & 3)] : 0.0;
        const int lf = cb + (lane & 15);
        skipv = (lf < c1) ? (out_of_bounds[lf] & 1) : 1;
    };
    if constexpr (KX > 0) {
        static_assert(KX == 0 || KMAX == 4, "one coefficient per lane");
        request_coefficients(c0, cf_next, skip_next);
    } else {
        stage_coefficients(c0, 0);
        block_sync<16>();
    }
    int buf = 0;
    for (int cb = c0; cb < c1; cb += 16, buf ^= 1) {
        if constexpr (KX > 0) {
            cf_lanes = cf_next;
            skip_lanes = skip_next;
            if (cb + 16 < c1)
                request_coefficients(cb + 16, cf_next, skip_next);
        }
        double part[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            double acc = 0.0;
            const double *cfp = s_cf + (buf * 16 + g) * CFS;
            bool wanted;
            if constexpr (KX > 0)
                wanted = __builtin_amdgcn_readlane(skip_lanes, g) == 0;
            else
                wanted = cfp[KMAX] == 0.0;                // (wave-uniform: an LDS broadcast)
            if (wanted) {
                // coefficients beyond K read the (zeroed) padding of the scratch row
                double cf[KMAX];
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    cf[k] = (KX > 0) ? ((k < KX) ? read_lane(cf_lanes, 4 * g + k) : 0.0) : cfp[k];
#pragma unroll
                for (int q = 0; q < TP; ++q) {
                    double m;
                    if constexpr (LINEAR) {
                        // columns k >= K are zero in registers and in cf: the extra
                        // fma(0, 0, m) leave m unchanged, no per-term guard needed
                        m = a[q][0] * cf[0];
#pragma unroll
                        for (int k = 1; k < KMAX; ++k)
                            m = mc3_fma(a[q][k], cf[k], m);
                    } else if constexpr (KX > 0) {
                        m = cf[KX - 1];           // Horner from coefficient K-1 down
#pragma unroll
                        for (int k = KX - 2; k >= 0; --k)
                            m = mc3_fma(m, a[q][0], cf[k]);
                    } else {
                        m = 0.0;                  // (the same with K known at run time only)
#pragma unroll
                        for (int k = KMAX - 1; k >= 0; --k) {
                            if (k == K - 1)
                                m = cf[k];
                            else if (k < K - 1)
                                m = mc3_fma(m, a[q][0], cf[k]);
                        }
                    }
                    const double r = (m - d[q]) * w[q];
                    acc = mc3_fma(r, r, acc);
                }
            }
            part[g] = acc;
        }
#pragma unroll
        for (int g = 0; g < 16; ++g)
            s_x[g * 1024 + l] = part[g];
        if constexpr (KX == 0) {
            if (cb + 16 < c1)
                stage_coefficients(cb + 16, buf ^ 1);
        }
        block_sync<16>();
        {
            const int lc = cb + wv;                     // wave wv reduces chain cb + wv
            double p[16];
#pragma unroll
            for (int w2 = 0; w2 < 16; ++w2)
                p[w2] = s_x[wv * 1024 + lane + 64 * w2];
#pragma unroll
            for (int s2 = 8; s2 >= 1; s2 >>= 1) {
#pragma unroll
                for (int w2 = 0; w2 < s2; ++w2)
                    p[w2] = p[w2] + p[w2 + s2];
            }
            const double tot = wave_sum(p[0]);
            if (lane == 0 && lc < c1)
                P.sc_part[(size_t)lc * P.ntiles + t] = tot;
        }
        block_sync<16>();
    }
}

// The same data sums through the FP64 matrix cores (design-matrix
// models only): model[chain][point] = sum_k coef[chain][k] * A[k][point] is a K <= 16 deep
// contraction, and v_mfma_f64_16x16x4_f64 computes it for 16 chains x 16 points at once.
// Measured on gfx950 (tools/probes/mfma_f64_probe.hip): the instruction adds its four
// products in k order with one rounding each -- exactly the fused-multiply-add chain of the
// vector kernel above, bit for bit -- occupies a SIMD's matrix pipe for 64 cycles (1024
// FMAs: the FP64 vector peak), and reaches that rate from a single wave.  What it buys:
// the coefficients are an ordinary per-lane operand (no broadcast through SGPRs or LDS,
// which is what stalls the vector kernel), and the VALU is free for the residual epilogue.
// Layout per instruction: A = coefficients (row = chain, lane&15; k = lane>>4), B = design
// matrix (k = lane>>4; column = point, lane&15), D rows (lane>>4) + 4 r = chains, column =
// point.  Wave wv owns virtual lanes [64 wv, 64 wv + 64) as four blocks of 16 points, each
// with its two points of the tile; design values, data and weights of those points stay
// in registers; results go through the same LDS array and the same tree as above.

template <int KB>
__global__ __launch_bounds__(512) void tile_chisq_mfma_kernel(
    const DevProblem P, const int chain_block, const double *__restrict__ nextp,
    const int32_t *__restrict__ out_of_bounds)
{
    // Summation order of the matrix-core kernels (mc3hip_plan.tile_tree = 2; the same in
    // resident.h ResidentTile): the weights are folded into the operand rows (P.aw: design
    // columns and data times w), so the contraction delivers the finished residual -- the data
    // ride as operand row K with coefficient -1 -- and the epilogue is ONE fused multiply-add
    // per value; a lane accumulates its sixteen points of the tile in step order (block vb,
    // half h), the 16 lanes of a row are added once per block of chains (strides 8..1), then
    // the tile's eight rows = waves (4..1).  Eight waves: the operand rows of a wave's 256
    // points stay in registers; the matrix pipe is busy from one wave per SIMD on.
    __shared__ double s_w[2][16][16];                  // [buffer][chain][row = wave; 8..15 zero]
    const int l = threadIdx.x, lane = l & 63, wv = l >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int t = blockIdx.x;
    const int K = P.nmodel;
    double bop[8][2][KB];
#pragma unroll
    for (int vb = 0; vb < 8; ++vb) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const size_t i = (size_t)t * 2048 + 1024 * h + 128 * wv + 16 * vb + j;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int k = 4 * kb + g;
                bop[vb][h][kb] = (k <= K) ? P.aw[(size_t)k * P.npad + i] : 0.0;
            }
        }
    }
    const int c0 = blockIdx.y * chain_block;
    const int c1 = (c0 + chain_block < P.nlocal) ? c0 + chain_block : P.nlocal;
    int buf = 0;
    for (int cb = c0; cb < c1; cb += 16, buf ^= 1) {
        // this lane's coefficient operand: chain cb + j, coefficients 4 kb + g (the scratch
        // rows are zero beyond K; position K is the data row's -1)
        double aop[KB];
        {
            const int lc = (cb + j < c1) ? cb + j : c1 - 1;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int k = 4 * kb + g;
                aop[kb] = (k == K) ? -1.0 : nextp[(size_t)lc * SC_STRIDE + k];
            }
        }
        double acc[4] = {0.0, 0.0, 0.0, 0.0};          // chain g + 4 r
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int vb = st >> 1, h = st & 1;
            mc3_v4d m = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                m = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[kb], bop[vb][h][kb], m, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[r] = mc3_fma(m[r], m[r], acc[r]);
        }
        // the 16 lanes of a row
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double s = add_ror<8>(acc[r]);
            s = add_ror<4>(s);
            s = add_ror<2>(s);
            s = add_ror<1>(s);
            const int lc = cb + g + 4 * r;
            const bool skip = lc >= c1 || (out_of_bounds[lc < c1 ? lc : c1 - 1] & 1);
            if (j == 0) {
                s_w[buf][g + 4 * r][wv] = skip ? 0.0 : s;
                s_w[buf][g + 4 * r][8 + wv] = 0.0;
            }
        }
        block_sync<8>();
        if (lane < 32) {                    // wave wv: the eight row totals of chains cb + 2 wv, + 1
            const int ch = 2 * wv + (lane >> 4);
            double s = s_w[buf][ch][lane & 15];
            s = add_ror<8>(s);              // (+ 0: slots 8..15)
            s = add_ror<4>(s);
            s = add_ror<2>(s);
            s = add_ror<1>(s);
            if ((lane & 15) == 0 && cb + ch < c1)
                P.sc_part[(size_t)(cb + ch) * P.ntiles + t] = s;
        }
        // (no second barrier: the next block writes the other buffer, and this one is not
        // written again before everybody has passed the next barrier)
    }
}

// tiled plan: proposal and Metropolis/accept halves of one iteration (one wave per chain)
template <int MT>
__global__ __launch_bounds__(64) void propose_kernel(const DevProblem P, const int roff,
                                                     const int j)
{
    run_chain<1, MT, false, false, false, -1, false, PH_PROPOSE>(P, P.ctl[0] + roff, 1, j, j + 1);
}

template <int MT>
__global__ __launch_bounds__(64) void finish_kernel(const DevProblem P, const int roff,
                                                    const int j)
{
    run_chain<1, MT, false, false, false, -1, false, PH_FINISH>(P, P.ctl[0] + roff, 1, j, j + 1);
    if (j + 1 == P.thinning)
        ring_out(P, roff);
}

}  // namespace mc3hip
