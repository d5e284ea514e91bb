// gauss.h -- Gaussian chi-squared data term of one chain (fused plan).
#pragma once
#include "core.h"

namespace mc3hip {

// Gaussian chi-squared data term, src_c/_chisq.c:131-133 with w = 1/uncert.
// Returns +inf when any model value is +inf (chain.py:322-323).
// (T1: the second 1024-point tile is register-resident too, in tile1)
template <int WPC, int MT, bool MULTI, bool T1 = false>
__device__ __forceinline__ double block_gauss_model(const DevProblem &P, const Model<MT> &model,
                                                    double *s_part, int *s_flag,
                                                    const DataTile<WPC, MT> &tile,
                                                    const DataTile<WPC, MT> *tile1 = nullptr);

// One lane's share of the data sum of a one-wave chain: the sixteen residuals of its virtual
// lanes and the in-lane levels of lane_tree<1> (the several-chains-at-once loops finish the
// tree across lanes for all their chains together, rows_sum).
template <int MT>
__device__ __forceinline__ double gauss_lane_share(const Model<MT> &model,
                                                   const DataTile<1, MT> &tile)
{
    static_assert(!mt_is_link(MT) && MT != MT_LINEAR, "polynomial families");
    double acc[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const double m = model.at_x(tile.x[v]);
        const double r = (m - tile.d[v]) * tile.w[v];
        acc[v] = mc3_fma(r, r, 0.0);
    }
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) {
#pragma unroll
        for (int v = 0; v < s; ++v)
            acc[v] = acc[v] + acc[v + s];
    }
    return acc[0];
}

template <int WPC, int MT, bool MULTI>
__device__ double block_gauss(const DevProblem &P, const double *s_nextp, double *s_part,
                              int *s_flag, const DataTile<WPC, MT> &tile)
{
    Model<MT> model;
    model.init(P, s_nextp);
    return block_gauss_model<WPC, MT, MULTI>(P, model, s_part, s_flag, tile);
}

template <int WPC, int MT, bool MULTI, bool T1>
__device__ __forceinline__ double block_gauss_model(const DevProblem &P, const Model<MT> &model,
                                                    double *s_part, int *s_flag,
                                                    const DataTile<WPC, MT> &tile,
                                                    const DataTile<WPC, MT> *tile1)
{
    static_assert(!mt_is_link(MT) || (!MULTI && !T1), "compile-time columns: one tile");
    constexpr int NV = 16 / WPC;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc[NV];
    const int npad = P.npad;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double m;
        if constexpr (mt_is_link(MT))
            m = model.from_columns(tile.a[v]);
        else if constexpr (MT == MT_LINEAR)
            m = model.at(lane + 64 * (wv + WPC * v));
        else
            m = model.at_x(tile.x[v]);
        const double r = (m - tile.d[v]) * tile.w[v];
        acc[v] = mc3_fma(r, r, 0.0);
    }
    if constexpr (T1) {                               // second tile, also in registers
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const double m = model.at_x(tile1->x[v]);
            const double r = (m - tile1->d[v]) * tile1->w[v];
            acc[v] = mc3_fma(r, r, acc[v]);
        }
    }
    // points beyond the register-resident tiles are streamed (MULTI = ndata > 1024)
    for (int base = (T1 ? 2 : 1) * VLANES; MULTI && base < npad; base += VLANES) {
        double xs[NV], ds[NV], ws[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {                // issue every load of the tile first
            const int i = base + lane + 64 * (wv + WPC * v);
            ds[v] = P.data[i];
            ws[v] = P.w[i];
            if constexpr (MT != MT_LINEAR)
                xs[v] = P.indep[i];
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            double m;
            if constexpr (MT == MT_LINEAR)
                m = model.at(base + lane + 64 * (wv + WPC * v));
            else
                m = model.at_x(xs[v]);
            const double r = (m - ds[v]) * ws[v];
            acc[v] = mc3_fma(r, r, acc[v]);
        }
    }
    // The "+inf model => chisq = inf" trap of chain.py:322-323 is resolved lazily: a
    // +inf model value makes its own partial sum non-finite (inf, or NaN where w = 0),
    // and a non-finite partial makes the total non-finite (inf + x, inf - inf, NaN + x),
    // so only a non-finite total has the model values scanned to tell a +inf model from
    // any other overflow.  The hot path pays one compare for it.
    const double total = lane_tree<WPC>(acc, s_part);
    const bool bad = threadIdx.x < 64 &&
                     (mc3_bits(total) & 0x7ff0000000000000ull) == 0x7ff0000000000000ull;
    if (__builtin_expect(block_or<WPC>(bad, s_flag), 0)) {
        int saw = 0;
        for (int i = threadIdx.x; i < P.ndata; i += 64 * WPC)
            saw |= is_pos_inf(model.at(i));
        if (block_or<WPC>(saw, s_flag))
            return mc3_from_bits(0x7ff0000000000000ull);
    }
    return total;
}

}  // namespace mc3hip
