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

// A thread's share of the data sum when WPC waves share a chain's (only) tile: the residuals of
// its 16 / WPC virtual lanes and the in-thread levels of lane_tree<WPC> (every model family).
// (NT: tiles of 1024 points the thread keeps a share of -- a virtual lane adds up its points
// tile after tile, the canonical order of mc3hip_plan.tile = 0; polynomial families)
template <int WPC, int MT, int NT = 1>
__device__ __forceinline__ double gauss_thread_share(const Model<MT> &model,
                                                     const DataTile<WPC, MT> (&tiles)[NT])
{
    static_assert(NT == 1 || MT != MT_LINEAR, "several tiles: the columns known at compile time");
    constexpr int NV = 16 / WPC;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const DataTile<WPC, MT> &tile = tiles[0];
    double acc[NV];
    if constexpr (NT > 1 && mt_is_link(MT)) {
        // (design matrices beyond one tile, round 5: column by column across the thread's
        // points of a tile, as below; tile after tile)
        constexpr int NC = Model<MT>::NC;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            double m[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v)
                m[v] = tiles[q].a[v][0] * model.cf[0];
#pragma unroll
            for (int k = 1; k < NC; ++k) {
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    m[v] = mc3_fma(tiles[q].a[v][k], model.cf[k], m[v]);
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const double r = (m[v] - tiles[q].d[v]) * tiles[q].w[v];
                acc[v] = mc3_fma(r, r, q == 0 ? 0.0 : acc[v]);
            }
        }
    } else if constexpr (NT > 1) {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const double m = model.at_x(tiles[q].x[v]);
                const double r = (m - tiles[q].d[v]) * tiles[q].w[v];
                acc[v] = mc3_fma(r, r, q == 0 ? 0.0 : acc[v]);
            }
        }
    } else if constexpr (mt_is_link(MT)) {
        // (column by column across the thread's points -- Model::from_columns' terms in its
        // order for every point, but the points' dependent chains side by side: written point
        // by point the compiler left half of them one fused multiply-add after the other)
        constexpr int NC = Model<MT>::NC;
        double m[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v)
            m[v] = tile.a[v][0] * model.cf[0];
#pragma unroll
        for (int k = 1; k < NC; ++k) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
                m[v] = mc3_fma(tile.a[v][k], model.cf[k], m[v]);
            // (... and kept so -- an empty asm every step's values pass through: instruction
            // selection, minding register pressure, put the chains back one after the other)
            if constexpr (NV == 4)
                asm volatile("; step" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]));
            else if constexpr (NV == 8)
                asm volatile("; step" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]),
                                        "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]));
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const double r = (m[v] - tile.d[v]) * tile.w[v];
            acc[v] = mc3_fma(r, r, 0.0);
        }
    } else {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            double m;
            if constexpr (MT == MT_LINEAR)
                m = model.at(lane + 64 * (wv + WPC * v));
            else
                m = model.at_x(tile.x[v]);
            const double r = (m - tile.d[v]) * tile.w[v];
            acc[v] = mc3_fma(r, r, 0.0);
        }
    }
#pragma unroll
    for (int s = NV / 2; s >= 1; s >>= 1) {
#pragma unroll
        for (int v = 0; v < s; ++v)
            acc[v] = acc[v] + acc[v + s];
    }
    return acc[0];
}

// Helper waves of a hand-laid-out one-wave loop: NW = 2 or 4 waves on different SIMDs of the CU
// share the data sum of the chain that wave 0 runs.  A design matrix of ten columns does not
// fit one wave's registers (16 points x 12 values per lane: half the tile sat in accumulation
// registers and every operand was moved before use); spread over four waves a thread keeps 4
// points x 12 values and the longest compute stretch of the iteration is a quarter as long.
// The helpers sleep at a barrier (A) until wave 0 has put the proposed coefficients in LDS,
// compute the sums of their virtual lanes (thread t owns virtual lanes t + 64 NW j: the
// canonical order of DataTile<NW>, so the result is bit for bit the one-wave loop's) and hand
// them back through LDS behind a second barrier (B).  Two raw barriers per iteration that
// evaluates (LDS traffic only: no vector-memory drain), none when the proposal is out of
// bounds; a waiting wave costs its SIMD nothing.  Who leaves is decided by wave 0 alone and
// said through LDS: a wave that left on its own would no longer be counted by the barrier the
// others wait at.
template <int MT, int NW, int NT = 1>
struct SumHelpers {
    static_assert(NW == 2 || NW == 4 || NW == 8, "2, 4 or 8 waves share the data sum");
    static constexpr int NC = Model<MT>::NC;
    double *s_part, *s_cf;
    int *s_cmd;                                  // [0]: 0 = a sum, 2 = leave; [1]: entry verdict

    // every wave, first thing: false = the launch is not to run (wave 0's reading of `stop`)
    __device__ __forceinline__ bool enter(double *part, double *cf, int *cmd, bool stop)
    {
        s_part = part;
        s_cf = cf;
        s_cmd = cmd;
        if (threadIdx.x == 0) {
            s_cmd[0] = 0;
            s_cmd[1] = stop ? 1 : 0;
        }
        block_sync<NW>();
        return s_cmd[1] == 0;
    }
    // waves 1..NW-1: until wave 0 says leave
    __device__ __forceinline__ void serve(const DevProblem &P,
                                          const DataTile<NW, MT> (&tile)[NT])
    {
        for (;;) {
            block_sync<NW>();                          // (A) the order of this iteration
            if (s_cmd[0] == 2)
                return;
            Model<MT> model;
            model.init(P, s_cf);
            s_part[threadIdx.x] = gauss_thread_share<NW, MT, NT>(model, tile);
            block_sync<NW>();                          // (B) the partial sums are in LDS
        }
    }
    // wave 0: the data sum of `model`, whose coefficient `lane` is cf_lane (lane < NC), in two
    // steps -- what only wave 0 computes (the prior term) belongs between them, beside the
    // helpers' work, not behind it
    __device__ __forceinline__ void order(double cf_lane, int lane)
    {
        if (lane < NC)
            s_cf[lane] = cf_lane;
        block_sync<NW>();                              // (A)
        // (nothing of this wave's share above the barrier: the scheduler is free to move pure
        // arithmetic across s_barrier, and did -- wave 0 then computed its share BEFORE waking
        // the helpers, i.e. the shares ran one after the other)
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ double collect(const DevProblem &P, const Model<MT> &model,
                                              const DataTile<NW, MT> (&tile)[NT], int lane)
    {
        const double t0 = gauss_thread_share<NW, MT, NT>(model, tile);
        block_sync<NW>();                              // (B)
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
        double total = wave_sum(p[0]);
        if (__builtin_expect((mc3_bits(total) & 0x7ff0000000000000ull) ==
                             0x7ff0000000000000ull, 0)) {
            // the "+inf model => chisq = inf" trap (chain.py:322-323), resolved lazily as in
            // block_gauss_model: only a non-finite total has the model values scanned
            int saw = 0;
            for (int i = lane; i < P.ndata; i += 64)
                saw |= is_pos_inf(model.at(i));
            if (__any(saw))
                total = mc3_from_bits(0x7ff0000000000000ull);
        }
        return total;
    }
    // wave 0, last thing
    __device__ __forceinline__ void release(int lane)
    {
        if (lane == 0)
            s_cmd[0] = 2;
        block_sync<NW>();                              // (A)
    }
};

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
