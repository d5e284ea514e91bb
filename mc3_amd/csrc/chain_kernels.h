// chain_kernels.h -- the fused chain-update kernels (gfx950, wave64).
//
// One workgroup advances one chain through `thinning` iterations of
//   proposal -> bounds -> shared params -> model -> chisq -> Metropolis -> Z append
// (mc3/chain.py:185-289) with every other chain seen through the population
// snapshot of the previous Z refresh.  WPC waves (64*WPC threads) share one chain:
// the data sum is split over 1024 "virtual lanes" (point i belongs to lane
// i mod 1024); a thread owns 16/WPC of them, so the summation order -- strided
// partial sums, then a halving tree over the 1024 lanes -- is the same for every
// WPC and is what oracle/oracle_lockstep.c follows.  No MFMA: with <= 10 model
// coefficients there is no dense contraction to feed it.
//
// Arithmetic contract: compiled with -ffp-contract=off; fused multiply-adds only
// where written (model, squared-residual accumulation, D4 taps, detmath).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "philox.h"
#include "../../include/mc3hip.h"
#include "../../include/mc3hip_detmath.h"

namespace mc3hip {

constexpr int VLANES = 1024;
// Z rows are padded on the device to whole 128-byte lines (ZROW doubles, or a multiple of
// it for more than 16 free parameters): rows are written by different chains -- different
// CUs and XCDs -- at the same moment, and write-through stores that share a line
// serialise on it.  Measured on C2: 165 -> 194 M chain-steps/s.  The C-ABI packs and
// unpacks, so callers keep seeing dense [rows][nfree] arrays.
constexpr int ZROW = 16;

// model tags
constexpr int MT_POLY_GENERIC = 0;   // runtime nmodel, coefficients read from LDS
constexpr int MT_POLY2 = 2;          // p0 + p1 x
constexpr int MT_POLY3 = 3;          // p0 + p1 x + p2 x^2
constexpr int MT_POLY4 = 4;
constexpr int MT_LINEAR = 100;       // sum_k A[k][i] p[k]

struct DevProblem {
    int32_t nchains, chain0, nlocal, nparams, nfree, nshare, ndata;
    int32_t npad;          // ndata rounded up to a multiple of 1024 (Gaussian data sum)
    int32_t zs, _pad_zs;   // doubles per row of Z and snap0 on the device (multiple of ZROW)
    int32_t sampler, wlike, nmodel, thinning, ngprior, trace_iters;
    int64_t M0;
    uint64_t seed;
    double gamma, fepsilon;
    const int32_t *ifree, *ishare, *share_src, *igprior;
    const double *params, *pmin, *pmax, *pstep, *prior, *priorlow, *priorup;
    const double *fmin, *fmax, *fstep;   // pmin/pmax/pstep gathered at ifree[k]
    const double *indep, *data, *w;
    double *Z, *log_post;
    int32_t *zchain;
    double *X, *chisq;
    long long *numaccept;
    unsigned long long *outbounds;
    double *best_chisq;
    long long *best_iter;
    double *bestx;
    const double *snap0;
    double *trace_chisq;
    int32_t *trace_flag;
    const long long *ctl;   // ctl[0] = refresh base of this replay, ctl[1] = first refresh of the run
    unsigned long long *stamps;   // diagnostic build only (MC3HIP_STAMPS): s_memtime per phase
    // peer-to-peer hand-off across GPUs (persistent kernel): rank p's replicas of Z and
    // log_post, mapped into this process; [0..npeers), entry `rank` is the local one
    int32_t npeers, rank;
    int32_t test_abort;           // tests: refresh at which chain `chain0` gives up (-1 = never)
    int32_t _pad_test;
    double *const *peerZ;
    double *const *peerLP;
    // tiled plan (large ndata): proposal / data sum / accept are separate launches
    int32_t tile, ntiles;         // points per tile (0 = fused plan), tiles in npad
    double *sc_nextp;             // [nlocal][nparams] proposed parameter vectors
    double *sc_npk;               // [nlocal][nfree]   proposed free parameters
    double *sc_scal;              // [nlocal][2]       Metropolis draw u, Hastings factor
    int32_t *sc_flag;             // [nlocal]          bit0 out of bounds, bit1 snooker jump
    double *sc_part;              // [nlocal][ntiles]  per-tile data sums
    // resident tiled plan (resident_tile_kernel): hand-off areas inside ONE launch, every
    // double its own "ready" flag (all-ones = not there yet, as for Z rows)
    double *rs_prop;              // [2][nlocal][RS_ROW] proposed coefficients, by iteration parity
    double *rs_part;              // [nlocal][rs_ntp]    per-tile data sums of the running iteration
    int32_t rs_ntp, _pad_rs;      // ntiles rounded up to whole 128-byte lines
    // wavelet likelihood, abscissae on np.linspace's grid (x_i = fl(fl(i*step) + x0), last
    // point = wl_xlast; checked bit for bit at creation): they are recomputed instead of
    // stored, which frees the LDS for the data values
    int32_t wl_grid, _pad_grid;
    double wl_x0, wl_step, wl_xlast;
    const double *data_t;         // wavelet: the data as [j][t] (thread t's j-th point)
    double *wl_scratch;           // wavelet, ndata > 2^14: [blocks][3 ndata / 4] pre-level values
    // RCCL exchange inside a hipGraph: slot `roff` of this ring receives the rows of the
    // refresh a graph node produced (fixed addresses, so the captured all-gather can be
    // replayed); nullptr = off
    double *xring;
};

constexpr int PH_FUSED = 0, PH_PROPOSE = 1, PH_FINISH = 2, PH_RESIDENT = 3;
constexpr int RS_ROW = 16;          // doubles per published proposal row: one 128-byte line
constexpr int RS_WCAP = 256;        // chains whose group totals a workgroup collects in LDS
                                    // before it adds them up (resident tiled plan)
constexpr int SC_STRIDE = MC3HIP_MAX_PARAMS + 16;   // doubles per sc_nextp row (zero padded)

// In-kernel phase stamps for the diagnostic build (never compiled into the product
// library): drain outstanding memory operations, then read the shader clock.
#ifdef MC3HIP_STAMPS
#define MC3_STAMP(slot)                                                                   \
    do {                                                                                  \
        unsigned long long t_;                                                            \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" \
                     : "=s"(t_)::"memory");                                               \
        if (threadIdx.x == 0 && blockIdx.x == 0)                                          \
            P.stamps[slot] = t_;                                                          \
    } while (0)
// non-draining tick accumulators for the persistent loop: sums of shader-clock
// deltas per phase over all iterations of workgroup 0 (slots 8..11), count in slot 12
#define MC3_TICK(var)                                                                     \
    do {                                                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");       \
    } while (0)
// Non-draining stamps for the kernels with dynamic LDS (wavelet path): the shader clock
// goes to a spare corner of the LDS noise table (slots 32..47 of `tab`, SmemLayout) and is
// copied out once per iteration (MC3_LSTAMP_FLUSH), so that no stamp waits for the store
// of the one before it.
extern __shared__ __attribute__((aligned(16))) double mc3_smem_stamps_[];
#ifdef MC3HIP_NO_LSTAMP
#define MC3_LSTAMP(slot) do { } while (0)
#define MC3_LSTAMP_FLUSH() do { } while (0)
#else
#define MC3_LSTAMP(slot)                                                                  \
    do {                                                                                  \
        /* (the builtin, not inline asm with an SGPR output: some stamps sit in control    \
           flow the compiler considers divergent) */                                      \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                       \
        if (threadIdx.x == 0 && blockIdx.x == 0)                                          \
            reinterpret_cast<unsigned long long *>(mc3_smem_stamps_)[blockDim.x +         \
                MC3HIP_MAX_PARAMS + 2 + 32 + (slot)] = t_;                                \
    } while (0)
#define MC3_LSTAMP_FLUSH()                                                                \
    do {                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x == 0)                                          \
            for (int q_ = 0; q_ < 16; ++q_)                                               \
                P.stamps[q_] = reinterpret_cast<unsigned long long *>(                    \
                    mc3_smem_stamps_)[blockDim.x + MC3HIP_MAX_PARAMS + 2 + 32 + q_];      \
    } while (0)
#endif
#else
#define MC3_STAMP(slot) do { } while (0)
#define MC3_TICK(var) do { } while (0)
#define MC3_LSTAMP(slot) do { } while (0)
#define MC3_LSTAMP_FLUSH() do { } while (0)
#endif

__device__ __forceinline__ double make_double(int lo, int hi) { return __hiloint2double(hi, lo); }

// ---- model evaluation ------------------------------------------------------
template <int MT>
struct Model {
    double cf[(MT >= 1 && MT <= 8) ? MT : 1];
    const double *coef;     // LDS copy of the proposal (generic / linear)
    const double *indep;
    int nmodel, ndata;      // ndata = column stride of A (the padded length)

    __device__ __forceinline__ void init(const DevProblem &P, const double *s_nextp)
    {
        coef = s_nextp;
        indep = P.indep;
        nmodel = P.nmodel;
        ndata = P.npad;
        if constexpr (MT >= 1 && MT <= 8) {
#pragma unroll
            for (int j = 0; j < MT; ++j)
                cf[j] = s_nextp[j];
        }
    }
    // coefficients straight from wave 0's lanes (lane j holds parameter j): the
    // fast path of fully-free polynomial models, no LDS round trip
    __device__ __forceinline__ void init_from_lanes(const DevProblem &P, double npk)
    {
        coef = nullptr;
        indep = P.indep;
        nmodel = P.nmodel;
        ndata = P.npad;
        if constexpr (MT >= 1 && MT <= 8) {
#pragma unroll
            for (int j = 0; j < MT; ++j)
                cf[j] = make_double(__builtin_amdgcn_readlane(__double2loint(npk), j),
                                    __builtin_amdgcn_readlane(__double2hiint(npk), j));
        }
    }
    // (lane k holds FREE parameter k; coefficient j is free parameter kf[j], or -- kf[j] < 0
    // -- the fixed value fx[j])
    __device__ __forceinline__ void init_from_lanes(const DevProblem &P, double npk,
                                                    const int (&kf)[MT >= 1 && MT <= 8 ? MT : 1],
                                                    const double (&fx)[MT >= 1 && MT <= 8 ? MT : 1])
    {
        coef = nullptr;
        indep = P.indep;
        nmodel = P.nmodel;
        ndata = P.npad;
        if constexpr (MT >= 1 && MT <= 8) {
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int src = kf[j] < 0 ? 0 : kf[j];
                const double v = make_double(__builtin_amdgcn_readlane(__double2loint(npk), src),
                                             __builtin_amdgcn_readlane(__double2hiint(npk), src));
                cf[j] = kf[j] < 0 ? fx[j] : v;
            }
        }
    }
    // polynomial families at abscissa x (Horner, fused multiply-adds)
    __device__ __forceinline__ double at_x(double x) const
    {
        if constexpr (MT >= 1 && MT <= 8) {
            double m = cf[MT - 1];
#pragma unroll
            for (int j = MT - 2; j >= 0; --j)
                m = mc3_fma(m, x, cf[j]);
            return m;
        } else {
            double m = coef[nmodel - 1];
            for (int j = nmodel - 2; j >= 0; --j)
                m = mc3_fma(m, x, coef[j]);
            return m;
        }
    }
    __device__ __forceinline__ double at(int i) const
    {
        if constexpr (MT != MT_LINEAR) {
            return at_x(indep[i]);
        } else {
            double m = indep[i] * coef[0];
            for (int k = 1; k < nmodel; ++k)
                m = mc3_fma(indep[(size_t)k * ndata + i], coef[k], m);
            return m;
        }
    }
};

__device__ __forceinline__ bool is_pos_inf(double v)
{
    return mc3_bits(v) == 0x7ff0000000000000ull;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains
// the vector-memory counter, which would serialise the data loads this kernel
// deliberately keeps in flight across the proposal phase; gfx950's s_barrier
// itself waits for no counter.  One wave per workgroup needs no barrier at all:
// a wave's LDS instructions execute in order.
template <int WPC>
__device__ __forceinline__ void block_sync()
{
    if constexpr (WPC == 1) {
        __builtin_amdgcn_wave_barrier();
    } else {
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0) only
        __builtin_amdgcn_s_barrier();
    }
}

template <int WPC>
__device__ __forceinline__ int block_or(int pred, int *s_flag)
{
    int any = __any(pred) ? 1 : 0;
    if constexpr (WPC > 1) {
        if (threadIdx.x == 0)
            s_flag[1] = 0;
        block_sync<WPC>();
        if (any && (threadIdx.x & 63) == 0)
            s_flag[1] = 1;
        block_sync<WPC>();
        any = s_flag[1];
    }
    return any;
}

// v[l] + v[l ^ 32], v[l] + v[l ^ 16]: gfx950 lane-swap instructions (VALU only)
__device__ __forceinline__ double add_xor32(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return make_double(a[0], b[0]) + make_double(a[1], b[1]);
}
__device__ __forceinline__ double add_xor16(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return make_double(a[0], b[0]) + make_double(a[1], b[1]);
}
// v[l] + v[(l + N) mod 16 within its row of 16]: DPP row rotate
template <int N>
__device__ __forceinline__ double add_ror(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    // (a rotation reads a valid lane everywhere: no "old" value to initialise)
    const int lo2 = __builtin_amdgcn_mov_dpp(lo, 0x120 + N, 0xf, 0xf, true);
    const int hi2 = __builtin_amdgcn_mov_dpp(hi, 0x120 + N, 0xf, 0xf, true);
    return v + make_double(lo2, hi2);
}
// sum over the 64 lanes of a wave, pairing strides 32,16,8,4,2,1; every lane
// receives the total
__device__ __forceinline__ double wave_sum(double t)
{
    t = add_xor32(t);
    t = add_xor16(t);
    t = add_ror<8>(t);
    t = add_ror<4>(t);
    t = add_ror<2>(t);
    t = add_ror<1>(t);
    return t;
}

__device__ __forceinline__ double read_lane(double v, int k)
{
    return make_double(__builtin_amdgcn_readlane(__double2loint(v), k),
                       __builtin_amdgcn_readlane(__double2hiint(v), k));
}

// Halving tree over the 1024 virtual lanes.  acc[v] is this thread's partial for
// virtual lane  lane + 64*(wave + WPC*v).  Strides >= 64*WPC are in-thread,
// strides 64*WPC/2..64 go through LDS, strides 32..1 are in-wave lane swaps.
// The result is valid in every lane of wave 0.
template <int WPC>
__device__ __forceinline__ double lane_tree(double (&acc)[16 / WPC], double *s_part)
{
    constexpr int NV = 16 / WPC;
#pragma unroll
    for (int s = NV / 2; s >= 1; s >>= 1) {
#pragma unroll
        for (int v = 0; v < s; ++v)
            acc[v] = acc[v] + acc[v + s];
    }
    double t = acc[0];
    if constexpr (WPC > 1) {
        block_sync<WPC>();                    // s_part may still be read by wave 0
        s_part[threadIdx.x] = t;
        block_sync<WPC>();
        if (threadIdx.x < 64) {
            double p[WPC];
#pragma unroll
            for (int w = 0; w < WPC; ++w)
                p[w] = s_part[threadIdx.x + 64 * w];
#pragma unroll
            for (int s = WPC / 2; s >= 1; s >>= 1) {
#pragma unroll
                for (int w = 0; w < s; ++w)
                    p[w] = p[w] + p[w + s];
            }
            t = p[0];
        }
    }
    if (threadIdx.x < 64)
        t = wave_sum(t);
    return t;
}

// The first 1024-point tile of (x, data, w) held in registers: loaded once at
// kernel entry -- before the proposal is known, so the load latency overlaps the
// RNG / partner-row phase -- and reused by every iteration of the launch.
template <int WPC, int MT>
struct DataTile {
    static constexpr int NV = 16 / WPC;
    double x[NV], d[NV], w[NV];

    // The device copies of x / data / w are padded to a multiple of 1024 points
    // (x and data repeat the last point, w = 0), so the tile needs no bounds
    // checks: a padded point adds fma(0, 0, acc) = acc.
    __device__ __forceinline__ void load(const DevProblem &P, int base = 0)
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = base + lane + 64 * (wv + WPC * v);
            d[v] = P.data[i];
            w[v] = P.w[i];
            if constexpr (MT != MT_LINEAR)
                x[v] = P.indep[i];
        }
    }
};

// Gaussian chi-squared data term, src_c/_chisq.c:131-133 with w = 1/uncert.
// Returns +inf when any model value is +inf (chain.py:322-323).
// (T1: the second 1024-point tile is register-resident too, in tile1)
template <int WPC, int MT, bool MULTI, bool T1 = false>
__device__ __forceinline__ double block_gauss_model(const DevProblem &P, const Model<MT> &model,
                                                    double *s_part, int *s_flag,
                                                    const DataTile<WPC, MT> &tile,
                                                    const DataTile<WPC, MT> *tile1 = nullptr);

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
    constexpr int NV = 16 / WPC;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc[NV];
    const int npad = P.npad;
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

constexpr double D4C0 = 0.4829629131445341, D4C1 = 0.83651630373780772,
                 D4C2 = 0.22414386804201339, D4C3 = -0.12940952255126034;

__device__ __forceinline__ void d4_pair(double a0, double a1, double a2, double a3,
                                        double &s, double &d)
{
    s = D4C0 * a0;
    s = mc3_fma(D4C1, a1, s);
    s = mc3_fma(D4C2, a2, s);
    s = mc3_fma(D4C3, a3, s);
    d = D4C3 * a0;
    d = mc3_fma(-D4C2, a1, d);
    d = mc3_fma(D4C1, a2, d);
    d = mc3_fma(-D4C0, a3, d);
}

__device__ __forceinline__ int ilog2(int n)
{
    int m = 0;
    while ((1 << m) < n)
        ++m;
    return m;
}

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
template <int NV, int CAP, int WPC, bool SYNC, int LEV, int NACC>
__device__ __forceinline__ void block_levels(double (&v)[CAP], double (&accs)[NACC],
                                             double2 *hb, int stride, int me, int next,
                                             bool active)
{
    if constexpr (NV >= 2) {
        static_assert(LEV < NACC, "one slot per level");
        double2 *h = hb + (LEV & 1) * stride;
        if (active)
            h[me] = make_double2(v[0], v[1]);
        if constexpr (SYNC)
            block_sync<WPC>();
        else
            __builtin_amdgcn_wave_barrier();
        const double2 hv = h[next];
        v[NV] = hv.x;
        v[NV + 1] = hv.y;
        accs[LEV] = block_level<NV, CAP>(v);
        block_levels<NV / 2, CAP, WPC, SYNC, LEV + 1, NACC>(v, accs, hb, stride, me, next, active);
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
    auto resid = [&](int j) {                           // j: static index into the block;
        double m, dv;                                   // B, B+1: the next block's first two
        if constexpr (VALUES) {
            return in0[j];
        } else if constexpr (MT == MT_LINEAR) {
            m = model.at(j < B ? base + j : (base + j) & (N - 1));
            dv = in0[j];
        } else if constexpr (GRID) {
            // numpy.linspace: y = arange(num) * step + start, y[-1] = stop
            const double fi = (j < B) ? fbase + (double)j : fnext + (double)(j - B);
            double x = fi * P.wl_step + P.wl_x0;
            if (j == B - 1)
                x = (me == Ta - 1) ? P.wl_xlast : x;
            if (j >= B && N <= 2)
                x = (j - B == N - 1) ? P.wl_xlast : x;
            m = model.at_x(x);
            dv = in0[j];
        } else {
            m = model.at_x(in0[j]);
            dv = in1[j];
        }
        sawinf |= is_pos_inf(m);
        return dv - m;
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
    } else {
        v[0] = resid(0);                                // N = 2: no level at all
        v[1] = resid(1);
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
    double term = 0.0;
    if (lane >= 1 && lane < M) {
        const double sW2 = sr2 * mc3_det_exp2(-gam * (double)lane) + sw2;
        s_tab[lane] = 1.0 / sW2;
        term = (double)(1 << lane) * mc3_det_log(6.283185307179586 * sW2);
    } else if (lane == 0) {
        const double sS2 = sr2 * mc3_det_exp2(-gam) * 0.72134752 + sw2;
        s_tab[24] = sS2;
        term = 2.0 * mc3_det_log(6.283185307179586 * sS2);
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
template <int WPC, int MT, int B, bool VALUES = false>
__device__ __forceinline__ double wavelet_body(const DevProblem &P, const double *s_nextp,
                                               double *s_part, int *s_flag, double *s_tab,
                                               double *s_halo, const double *xs,
                                               const double *values = nullptr, int npre = 0,
                                               const double *accP = nullptr, int sawinf0 = 0)
{
    constexpr int T = 64 * WPC;
    constexpr int NLA = ilog2c(B);                      // levels of stage A (N >= 4)
    constexpr int NLB = ilog2c(WPC);                    // levels of stage B
    const int tid = threadIdx.x, lane = tid & 63;
    // (made wave-uniform by hand so that the per-wave roles below are scalar branches)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = ilog2(P.ndata);
    const int N = VALUES ? B * T : P.ndata;             // length of the array stage A starts from
    const int Ta = N / B;                               // threads that own a block
    const bool active = tid < Ta;
    const int me = active ? tid : 0;
    const int next = (me + 1 == Ta) ? 0 : me + 1;
    Model<MT> model;
    model.init(P, s_nextp);
    if (WPC > 1 && tid == 0)
        s_flag[1] = 0;                                  // "a model value was +inf"
    // One wave: the noise terms first.  Several waves: the LAST wave computes them after its
    // share of stage A, while wave 0 is busy with stages B and C -- two long dependent chains
    // (software exp2 / log; cross-lane levels) side by side instead of one after the other.
    if constexpr (WPC == 1)
        wavelet_noise_terms(s_nextp, P.nparams, M, s_tab);
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
    else if (MT != MT_LINEAR && P.wl_grid)
        accA[0] = wavelet_level0<WPC, MT, B, true, HB + 2>(P, model, xs, me, next, Ta, sawinf, v);
    else
        accA[0] = wavelet_level0<WPC, MT, B, false, HB + 2>(P, model, xs, me, next, Ta, sawinf, v);
    MC3_LSTAMP(14);
    double2 *halo = reinterpret_cast<double2 *>(s_halo);  // [2][T]
    if (N >= 4) {
        block_levels<B / 2, HB + 2, WPC, true, 1, (NLA > 0 ? NLA : 1)>(v, accA, halo, T, me, next,
                                                                        active);
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
    int mscale = (N >= 4) ? mA - NLA : mA;              // scale of the next level's details
    double accB[NLB > 0 ? NLB : 1];
    if constexpr (WPC > 1) {
        // (a buffer of its own: the last halo exchange may still be being read by a wave
        // that is behind)
        double *ubuf = s_halo + 4 * T + 64;             // [T]
        ubuf[tid] = v[0];
        if (sawinf)
            s_flag[1] = 1;
        block_sync<WPC>();                              // (1) the T values are in LDS
        if (wave == WPC - 1)
            wavelet_noise_terms(s_nextp, P.nparams, M, s_tab);
        if (wave == 0) {
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
    if (wave == 0) {
        double *cbuf = s_halo + 4 * T;                  // [64], wave-private
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            if (nn >= 4) {
                const int nh = nn >> 1;
                __builtin_amdgcn_wave_barrier();
                cbuf[lane] = u;
                __builtin_amdgcn_wave_barrier();
                const int i2 = (lane < nh) ? 2 * lane : 0;
                const double2 ca = *reinterpret_cast<const double2 *>(cbuf + i2);
                const double2 cc = *reinterpret_cast<const double2 *>(cbuf + ((i2 + 2) & (nn - 1)));
                double s, d;
                d4_pair(ca.x, ca.y, cc.x, cc.y, s, d);
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
        block_sync<WPC>();                              // (2) the noise terms are in LDS
        s_part[tid] = (active && N >= 4) ? stage_a_sum() : 0.0;
        block_sync<WPC>();                              // (3) every thread's stage-A sum
        if (wave != 0)
            return 0.0;                                 // the value is only used by wave 0
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
            d4_pair(da.x - m0, da.y - m1, db.x - m2, db.y - m3, s2, d);
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
template <int WPC, int MT, bool LARGE = false>
__device__ double block_wavelet(const DevProblem &P, const double *s_nextp, double *s_part,
                                int *s_flag, double *s_tab, double *s_halo, const double *xs)
{
    constexpr int T = 64 * WPC;
    if constexpr (LARGE) {
        static_assert(WPC == 8, "the large-data path has eight waves");
        return wavelet_large<MT>(P, s_nextp, s_part, s_flag, s_tab, s_halo, xs);
    }
    const int B = (P.ndata / 2 < T) ? 2 : P.ndata / T;
#define MC3_WL_BODY(BB)                                                                       \
    return wavelet_body<WPC, MT, BB>(P, s_nextp, s_part, s_flag, s_tab, s_halo, xs)
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

// prior term, mc3/stats/stats.py:211-216 + src_c/include/stats.h:96-106
__device__ __forceinline__ double prior_term(const DevProblem &P, const double *s_nextp)
{
    double pr = 0.0;
    for (int g = 0; g < P.ngprior; ++g) {
        const int j = P.igprior[g];
        const double d = s_nextp[j] - P.prior[j];
        const double s = (d > 0.0) ? d / P.priorup[j] : d / P.priorlow[j];
        pr = pr + s * s;
    }
    return pr;
}

// chisq of the parameter vector in s_nextp; valid in wave 0
template <int WPC, int MT, bool WL, bool MULTI>
__device__ __forceinline__ double block_chisq(const DevProblem &P, const double *s_nextp,
                                              double *s_part, int *s_flag, double *s_tab,
                                              double *s_halo, const double *s_xs,
                                              const DataTile<WPC, MT> &tile)
{
    double c;
    if constexpr (WL)       // (MULTI of a wavelet kernel = its large-data instance)
        c = block_wavelet<WPC, MT, MULTI && WPC == 8>(P, s_nextp, s_part, s_flag, s_tab, s_halo,
                                                      s_xs);
    else
        c = block_gauss<WPC, MT, MULTI>(P, s_nextp, s_part, s_flag, tile);
    if (is_pos_inf(c))
        return c;
    return c + prior_term(P, s_nextp);
}

// serial sum over lanes 0..n-1 of a wave, in lane order (v_readlane broadcasts)
__device__ __forceinline__ double lane_serial_sum(double t, int n)
{
    double s = 0.0;
    for (int k = 0; k < n; ++k)
        s = s + read_lane(t, k);
    return s;
}

struct SmemLayout {
    double *part, *nextp, *tab, *halo, *xs;
    int *flag;
};

__device__ __forceinline__ SmemLayout carve(double *smem, const DevProblem &P, int nthreads)
{
    SmemLayout L;
    L.part = smem;
    L.nextp = smem + nthreads;
    L.flag = reinterpret_cast<int *>(L.nextp + MC3HIP_MAX_PARAMS);
    L.tab = L.nextp + MC3HIP_MAX_PARAMS + 2;            // 48 doubles: per-scale 1/sigma_W^2, ...
    L.halo = L.tab + 48;       // wavelet: two halo buffers of 2 doubles per thread, 64 for
    L.xs = L.halo + 5 * nthreads + 64;     // stage C, one value per thread; then x as [B][T]
    return L;
}

static inline size_t smem_bytes(int nthreads, int ndata, int wlike)
{
    size_t n = (size_t)nthreads + MC3HIP_MAX_PARAMS + 2 + 48;
    // (beyond 2^14 points nothing is staged in LDS: the pre-levels read global memory)
    n += wlike ? 5 * (size_t)nthreads + 64 + (size_t)(ndata > 16384 ? 128 : ndata > 128 ? ndata : 128) : 4;
    return n * sizeof(double);
}

// ---- the chain update -------------------------------------------------------
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
// the finish kernel adds the tiles in order.
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
typedef double mc3_v4d __attribute__((ext_vector_type(4)));

template <int KB>
__global__ __launch_bounds__(512) void tile_chisq_mfma_kernel(
    const DevProblem P, const int chain_block, const double *__restrict__ nextp,
    const int32_t *__restrict__ out_of_bounds)
{
    // Summation order of this kernel ("waves first", mc3hip_plan.tile_tree = 1): the 1024
    // virtual lanes of a tile are summed 64 at a time (halving tree, strides 32..1), then
    // the 16 group totals (strides 8..1).  A wave then hands over a few numbers per block of
    // chains instead of 16 x 64, and the 128 KiB exchange (and the matrix pipe idling behind
    // it) of the lanes-first order is gone.  Eight waves (two groups of 64 virtual lanes
    // each): the design values, data and weights of a wave's 256 points are 160 registers
    // per lane, which the 128 of a 1024-thread group cannot hold; the matrix pipe is busy
    // from one wave per SIMD on.
    __shared__ double s_w[2][16][16];                  // [buffer][chain][group of 64 lanes]
    const int l = threadIdx.x, lane = l & 63, wv = l >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int t = blockIdx.x;
    const int K = P.nmodel;
    double bop[8][2][KB], dd[8][2], ww[8][2];
#pragma unroll
    for (int vb = 0; vb < 8; ++vb) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const size_t i = (size_t)t * 2048 + 1024 * h + 128 * wv + 16 * vb + j;
            dd[vb][h] = P.data[i];
            ww[vb][h] = P.w[i];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int k = 4 * kb + g;
                bop[vb][h][kb] = (k < K) ? P.indep[(size_t)k * P.npad + i] : 0.0;
            }
        }
    }
    const int c0 = blockIdx.y * chain_block;
    const int c1 = (c0 + chain_block < P.nlocal) ? c0 + chain_block : P.nlocal;
    int buf = 0;
    for (int cb = c0; cb < c1; cb += 16, buf ^= 1) {
        // this lane's coefficient operand: chain cb + j, coefficients 4 kb + g (the scratch
        // rows are zero beyond K)
        double aop[KB];
        {
            const int lc = (cb + j < c1) ? cb + j : c1 - 1;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                aop[kb] = nextp[(size_t)lc * SC_STRIDE + 4 * kb + g];
        }
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            double acc[4][4];                          // [block of 16 lanes][chain g + 4 r]
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int vb = 4 * grp + q;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    mc3_v4d m = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
                        m = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[kb], bop[vb][h][kb], m, 0, 0,
                                                                  0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double res = (m[r] - dd[vb][h]) * ww[vb][h];
                        acc[q][r] = mc3_fma(res, res, h ? acc[q][r] : 0.0);
                    }
                }
            }
            // the group's 64 lanes (lane 16 q + j): strides 32 and 16 are in-thread, 8..1
            // run inside the 16-lane row
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double s = (acc[0][r] + acc[2][r]) + (acc[1][r] + acc[3][r]);
                s = add_ror<8>(s);
                s = add_ror<4>(s);
                s = add_ror<2>(s);
                s = add_ror<1>(s);
                const int lc = cb + g + 4 * r;
                const bool skip = lc >= c1 || (out_of_bounds[lc < c1 ? lc : c1 - 1] & 1);
                if (j == 0)
                    s_w[buf][g + 4 * r][2 * wv + grp] = skip ? 0.0 : s;
            }
        }
        block_sync<8>();
        if (lane < 32) {                    // wave wv: the 16 group totals of chains cb + 2 wv, + 1
            const int ch = 2 * wv + (lane >> 4);
            double s = s_w[buf][ch][lane & 15];
            s = add_ror<8>(s);
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

// chisq of a chain from the per-tile sums (tiles added in order) plus the prior
// term.  The +inf-model trap of chain.py:322-323 is resolved lazily: a +inf model
// value always makes the sum non-finite (w > 0), so only a non-finite sum needs
// the scan that tells "+inf model" (chisq = inf) from any other overflow.
template <int MT>
__device__ __forceinline__ double finish_chisq(const DevProblem &P, int lc)
{
    const double *np = P.sc_nextp + (size_t)lc * SC_STRIDE;
    double total = 0.0;
    for (int t = 0; t < P.ntiles; ++t)
        total = total + P.sc_part[(size_t)lc * P.ntiles + t];
    const bool finite = (mc3_bits(total) & 0x7ff0000000000000ull) != 0x7ff0000000000000ull;
    if (!finite) {
        Model<MT> model;
        model.init(P, np);
        int saw = 0;
        for (int i = threadIdx.x; i < P.ndata; i += blockDim.x)
            saw |= is_pos_inf(model.at(i));
        if (__any(saw))
            return mc3_from_bits(0x7ff0000000000000ull);
    }
    if (is_pos_inf(total))
        return total;
    return total + prior_term(P, np);
}

// A Z row read through the in-launch hand-off (persistent kernel): rows are
// published with agent-scope 8-byte stores and consumed with agent-scope loads
// (sc1: they bypass the CU's L1, which another CU's stores never refresh); a row
// that is not there yet still holds the all-ones NaN pattern the Z buffer is
// filled with at creation, so every double is its own "ready" flag -- no separate
// flag, fence or ordering is needed (8-byte stores are single-copy atomic).
// A chain state can never legitimately be NaN: a NaN proposal has a NaN or
// out-of-bounds chisq and is rejected (chain.py:257-260).
constexpr unsigned long long ROW_PENDING = 0xffffffffffffffffull;
constexpr int CTL_ABORT = 2;                   // ctl[2] != 0: a wait timed out

template <bool SYS = false>
__device__ __forceinline__ double load_published(const double *p)
{
    // SYS: the row may have been stored by another GPU (system scope, fine-grained
    // memory); otherwise by another CU of this GPU (agent scope)
    if constexpr (SYS)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wave-level wait until every lane's values have been published (all 64 lanes
// poll; lanes beyond nfree repeat lane 0's addresses).  The first read is issued
// by the caller so that independent scalar work can overlap its latency; this
// only spins when something is still pending.  Bounded: gives up (and tells
// everybody) after a few seconds so a lost workgroup can never hang the device;
// the host then restores its checkpoint and redoes the run (mc3hip.hip settle()).
template <int NVAL, bool SYS = false>
__device__ __forceinline__ bool wait_rows(const double *const (&addr)[NVAL], double (&val)[NVAL],
                                          long long *ctl, unsigned long long *nspin = nullptr)
{
    for (unsigned spins = 0;; ++spins) {
        if (nspin && spins)
            *nspin += 1;                      // diagnostic build: polls beyond the first
        bool pend[NVAL];
        bool any_pending = false;
#pragma unroll
        for (int q = 0; q < NVAL; ++q) {
            pend[q] = __any(mc3_bits(val[q]) == ROW_PENDING);
            any_pending |= pend[q];
        }
        if (!any_pending)
            return true;
        // (no pause, and only the rows still missing are asked for again: every extra request
        // on the fabric and every cycle of back-off measurably slows the hand-off)
        if (__builtin_expect((spins & 1023u) == 1023u, 0)) {
            // (made wave-uniform by hand: a per-lane load would make this exit, and with it
            // every value live across the chain loop, divergent in the compiler's eyes)
            const int ab = __builtin_amdgcn_readfirstlane(
                (int)(__hip_atomic_load(&ctl[CTL_ABORT], __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT) != 0));
            if (ab != 0 || spins > (1u << 23)) {
                // every lane stores the same word: a lane-0 branch right at this exit would
                // again make the exit look divergent
                if (ab == 0)
                    __hip_atomic_store(&ctl[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
#pragma unroll
        for (int q = 0; q < NVAL; ++q)
            if (pend[q])
                val[q] = load_published<SYS>(addr[q]);
    }
}

// The same wait with K polls kept in flight (in-order return: the oldest is checked while
// the younger ones travel), so that memory is sampled every round trip / K.
template <int K, int SYS>
__device__ __forceinline__ bool poll_ring(const double *const (&addr)[2], double (&v)[K][2],
                                          double (&val)[2], long long *ctl,
                                          unsigned long long *nspin = nullptr)
{
    for (unsigned spins = 0;; ++spins) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const bool pend0 = __any(mc3_bits(v[q][0]) == ROW_PENDING);
            const bool pend1 = __any(mc3_bits(v[q][1]) == ROW_PENDING);
            if (!pend0 && !pend1) {
                val[0] = v[q][0];
                val[1] = v[q][1];
                return true;
            }
            if (nspin)
                *nspin += 1;
            // only what is still missing is asked for again: every extra request on the
            // fabric slows everybody's hand-off (K = 2, 3, 4 polls in flight: 181, 173,
            // 140 M chain-steps/s against 197 for one)
            // (asked again at once: a pause of 64 / 128 / 256 / 512 cycles here measured
            // 197 / 195 / 191 / 182 M chain-steps/s against 205 without)
            if (K > 1 || pend0)
                v[q][0] = load_published<SYS>(addr[0]);
            if (K > 1 || pend1)
                v[q][1] = load_published<SYS>(addr[1]);
        }
        if (__builtin_expect((spins & 255u) == 255u, 0)) {
            const int ab = __builtin_amdgcn_readfirstlane(
                (int)(__hip_atomic_load(&ctl[CTL_ABORT], __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT) != 0));
            if (ab != 0 || spins > (1u << 22)) {
                if (ab == 0)
                    __hip_atomic_store(&ctl[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

// Keep a wave-uniform value where it was computed: an empty asm that "uses" it, so
// neither the IR-level sinking pass nor the scheduler can move the (scalar) Philox
// arithmetic of the next iteration out of the shadow of the partner-row loads.
__device__ __forceinline__ void pin_scalar(uint64_t &q)
{
    asm volatile("; pin" : "+s"(q));
}

// the wave-uniform draws of one iteration (slots 0..2)
struct Draws {
    Draw2 d0, d1, d2;
};

__device__ __forceinline__ Draws make_draws(const DevProblem &P, uint64_t iter, int c, int sampler)
{
    Draws D;
    D.d0 = draw2(P.seed, iter, (uint32_t)c, 0u);
    D.d1 = D.d0;
    D.d2 = D.d0;
    if (sampler == MC3HIP_SNOOKER) {
        D.d1 = draw2(P.seed, iter, (uint32_t)c, 1u);
        D.d2 = draw2(P.seed, iter, (uint32_t)c, 2u);
    }
    return D;
}

// The same draws for 21 consecutive iterations at once, on the vector unit: lane
// 3*(iter - base) + slot holds the four words of (iter, chain, slot).  The scalar Philox
// of make_draws costs a wave ~350 issue slots per call; it is free only while the wave
// waits for memory anyway.  Where nothing is left to wait for (persistent snooker, whose
// rows are requested an iteration ahead) a batch costs ~50 cycles per iteration instead.
struct DrawBatch {
    uint32_t w0, w1, w2, w3;
    long long base;
};
constexpr int DRAW_BATCH = 21;

__device__ __forceinline__ void fill_draws(DrawBatch &B, const DevProblem &P, long long iter0,
                                           int c)
{
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t it = (uint64_t)iter0 + (uint64_t)(lane / 3u);
    uint32_t w[4];
    philox4x32_10((uint32_t)it, (uint32_t)(it >> 32), (uint32_t)c, lane % 3u, (uint32_t)P.seed,
                  (uint32_t)(P.seed >> 32), w);
    B.w0 = w[0];
    B.w1 = w[1];
    B.w2 = w[2];
    B.w3 = w[3];
    B.base = iter0;
}

__device__ __forceinline__ Draw2 batch_draw(const DrawBatch &B, int l)
{
    Draw2 d;
    d.q0 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)B.w0, l) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)B.w1, l) << 32);
    d.q1 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)B.w2, l) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)B.w3, l) << 32);
    return d;
}

__device__ __forceinline__ Draws get_draws(DrawBatch &B, const DevProblem &P, long long iter,
                                           int c)
{
    if (iter < B.base || iter >= B.base + DRAW_BATCH)
        fill_draws(B, P, iter, c);
    const int l = 3 * (int)(iter - B.base);
    Draws D;
    D.d0 = batch_draw(B, l);
    D.d1 = batch_draw(B, l + 1);
    D.d2 = batch_draw(B, l + 2);
    return D;
}

// coefficient j of a fully-polynomial parameter vector: the free parameter it is (kf[j]) or
// its fixed value (fx[j], kf[j] = -1); wave-uniform, computed once per launch
template <int MT>
__device__ __forceinline__ void coefficient_map(const DevProblem &P, int (&kf)[MT], double (&fx)[MT])
{
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        kf[j] = -1;
        fx[j] = P.params[j];
    }
    for (int k = 0; k < P.nfree; ++k) {
        const int j = P.ifree[k];
#pragma unroll
        for (int q = 0; q < MT; ++q)
            kf[q] = (q == j) ? k : kf[q];
    }
    // shared parameters copy their source, in order (chain.py:246-247)
    for (int sh = 0; sh < P.nshare; ++sh) {
        const int j = P.ishare[sh], src = P.share_src[sh];
        int k = -1;
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < MT; ++q) {
            k = (q == src) ? kf[q] : k;
            v = (q == src) ? fx[q] : v;
        }
#pragma unroll
        for (int q = 0; q < MT; ++q) {
            kf[q] = (q == j) ? k : kf[q];
            fx[q] = (q == j) ? v : fx[q];
        }
    }
}

// prior_term for the lean FX instances: the proposal is in the lanes (free parameter k in
// lane k), not in LDS; same terms in the same order
template <int MT>
__device__ __forceinline__ double prior_from_lanes(const DevProblem &P, double npk,
                                                   const int (&kf)[MT], const double (&fx)[MT])
{
    double pr = 0.0;
    for (int g = 0; g < P.ngprior; ++g) {
        const int j = P.igprior[g];
        int k = -1;
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < MT; ++q) {
            k = (q == j) ? kf[q] : k;
            v = (q == j) ? fx[q] : v;
        }
        if (k >= 0)
            v = make_double(__builtin_amdgcn_readlane(__double2loint(npk), k),
                            __builtin_amdgcn_readlane(__double2hiint(npk), k));
        const double d = v - P.prior[j];
        const double s = (d > 0.0) ? d / P.priorup[j] : d / P.priorlow[j];
        pr = pr + s * s;
    }
    return pr;
}

// DEMC partners, chain.py:221-232: r1 != c, r2 != c, r2 != r1, from the two 32-bit
// halves of one draw
__device__ __forceinline__ void demc_partners(uint64_t q0, int n, int c, int &r1, int &r2)
{
    r1 = 1 + (int)bounded32((uint32_t)(q0 >> 32), (uint32_t)(n - 1));
    if (r1 == c)
        r1 = 0;
    r2 = r1 + 2 + (int)bounded32((uint32_t)q0, (uint32_t)(n - 2));   // < 2n
    r2 = (r2 >= n) ? r2 - n : r2;                                      // (...) % n
    if (r2 == c)
        r2 = (r1 + 1 == n) ? 0 : r1 + 1;                               // (r1+1) % n
}

// ---- resident tiled plan --------------------------------------------------------------
// The tiled plan as ONE launch (design-matrix models through the matrix cores; one GPU, or
// one launch per GPU with the Z rows published to every rank).  The grid is the tile kernel's
// -- tiles x columns of chains, one workgroup per CU -- but every workgroup keeps its tile
// (operand rows in registers and LDS, weights in LDS) for the whole launch instead of
// reloading 200 KB of it per iteration, and the three phases of an iteration hand over
// inside the launch instead of at kernel boundaries:
//   * a wave that OWNS a chain (wave v of workgroup w owns chain w + v * workgroups) runs the
//     ordinary run_chain for it -- proposal, Metropolis, Z append -- and publishes the proposed
//     coefficient row (rs_prop, one 128-byte line, by iteration parity);
//   * all eight waves of every workgroup then compute the data sums of their tile for the
//     chains of their column, 16 chains at a time, as soon as those rows are there, each wave
//     at its own pace; after the last block one barrier, the sums of the waves are added and
//     one sum per (chain, tile) is published (rs_part);
//   * the owner adds the sums of its chain in tile order, hands the slots and the consumed row
//     back (all-ones), and goes on to the next iteration.
// Every handed-over double is its own flag (all-ones = not there yet), exactly as for Z rows;
// no grid barrier, no fence.  Why a row may be reused: a slot is handed back only after all
// the sums that were computed from it have arrived, and a workgroup asks for the rows of
// iteration i + 2 only after it has used those of i + 1, which the owner published after
// handing back those of i.  The data column rides in the contraction as operand row K with
// coefficient -1 (D = model - data, the same single rounding as the subtraction it replaces).
// Same arithmetic and the same summation order as tile_chisq_mfma_kernel: the oracle does not
// know which of the two ran.
// What the owner of a chain does around the tile phase, whatever computes the tile sums
// (ResidentTile: design matrices through the matrix cores; ResidentPoly: polynomials on the
// vector unit).
struct ResidentOwner {
    int lane;
    int par;                // parity of the iteration whose proposals are used next
    bool failed;            // a wait of this wave timed out: said at the next agreement point
    // owner waves: their LDS area ([0..63]: the Gaussian priors as index, centre, lower and
    // upper width, 16 each)
    double *s_own;

    // this wave owns chain lc; `area` is its run_chain LDS area (whose first 64 doubles
    // run_chain does not use here)
    __device__ __forceinline__ void set_owner(const DevProblem &P, int lc, double *area)
    {
        s_own = area;
        if (P.ngprior <= 16 && lane < P.ngprior) {
            const int jp = P.igprior[lane];
            area[lane] = (double)jp;
            area[16 + lane] = P.prior[jp];
            area[32 + lane] = P.priorlow[jp];
            area[48 + lane] = P.priorup[jp];
        }
    }

    // owner, before the tile phase: the proposed coefficients of chain lc for everybody
    __device__ __forceinline__ void publish(const DevProblem &P, int lc, const double *s_nextp)
    {
        // (the slots handed back during the previous iteration have landed before the row
        // that lets anybody write them again leaves)
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
        if (lane < RS_ROW) {
            const int K = P.nmodel;
            double v = (lane < K) ? s_nextp[lane] : ((lane == K) ? -1.0 : 0.0);
            if (mc3_bits(v) == ROW_PENDING)
                v = mc3_from_bits(0x7ff8000000000000ull);
            __hip_atomic_store(&P.rs_prop[((size_t)par * P.nlocal + lc) * RS_ROW + lane], v,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // owner, after the tile phase: the data sum of chain lc, tiles added in order; hands the
    // slots and the row this iteration used back
    __device__ __forceinline__ double collect(const DevProblem &P, int lc)
    {
        long long *ctl = const_cast<long long *>(P.ctl);
        double *row = P.rs_part + (size_t)lc * P.rs_ntp;
        double total = 0.0;
        for (int t0 = 0; t0 < P.ntiles; t0 += 64) {
            const int n = (P.ntiles - t0 < 64) ? P.ntiles - t0 : 64;
            double *mine = row + t0 + (lane < n ? lane : 0);
            const double *const addr[1] = {mine};
            double val[1] = {load_published(mine)};
            if (!wait_rows<1>(addr, val, ctl))
                failed = true;
            // (in order, eight broadcasts at a time ahead of their dependent additions)
            int k = 0;
            for (; k + 8 <= n; k += 8) {
                double v8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v8[u] = read_lane(val[0], k + u);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    total = total + v8[u];
            }
            for (; k < n; ++k)
                total = total + read_lane(val[0], k);
            if (lane < n)
                __hip_atomic_store(mine, mc3_from_bits(ROW_PENDING), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane < RS_ROW)
            __hip_atomic_store(&P.rs_prop[((size_t)(par ^ 1) * P.nlocal + lc) * RS_ROW + lane],
                               mc3_from_bits(ROW_PENDING), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        return total;
    }

    // the chisq of the proposal from that sum (finish_chisq of the three-launch plan)
    template <int MT>
    __device__ __forceinline__ double finish(const DevProblem &P, double total,
                                             const double *s_nextp) const
    {
        const bool finite = (mc3_bits(total) & 0x7ff0000000000000ull) != 0x7ff0000000000000ull;
        if (!finite) {
            Model<MT> model;
            model.init(P, s_nextp);
            int saw = 0;
            for (int i = lane; i < P.ndata; i += 64)
                saw |= is_pos_inf(model.at(i));
            if (__any(saw))
                return mc3_from_bits(0x7ff0000000000000ull);
        }
        if (is_pos_inf(total))
            return total;
        if (P.ngprior > 16)
            return total + prior_term(P, s_nextp);
        // prior_term from the LDS copy of its constants: lane g works out term g (the
        // divisions side by side instead of one after the other), then the terms are added
        // in order
        double term = 0.0;
        if (lane < P.ngprior) {
            const int jp = (int)s_own[lane];
            const double d = s_nextp[jp] - s_own[16 + lane];
            const double sd = (d > 0.0) ? d / s_own[48 + lane] : d / s_own[32 + lane];
            term = sd * sd;
        }
        double pr = 0.0;
        for (int gp = 0; gp < P.ngprior; ++gp)
            pr = pr + read_lane(term, gp);
        return total + pr;
    }
};

template <int KB>
struct ResidentTile : ResidentOwner {
    // operand rows of the tile: blocks of four rows in registers, except -- with three of
    // them -- the last, which lives in LDS (64 KB; one read per product): 96 registers of
    // tile plus the products in flight plus the chain state of an owner wave do not fit 256
    static constexpr int KR = (KB == 3) ? 2 : KB;
    double bop[8][2][KR];
    const double *s_b;      // LDS [16][512]: this thread's operands of row block KR
    double *s_w;            // LDS [RS_WCAP][16]: group totals of the chains done so far
    const double *s_ww;     // LDS [2048]: the weights of the tile's points (this wave's: + 128 wv)
    int *s_stop;            // LDS: a wave of this workgroup has given up
    int t, c0, c1, wv;

    __device__ __forceinline__ void load(const DevProblem &P, int tile, int col, int chain_block,
                                         double *lds_w, int *lds_stop, double *lds_ww,
                                         double *lds_b)
    {
        s_b = lds_b + threadIdx.x;
        s_w = lds_w;
        s_stop = lds_stop;
        t = tile;
        c0 = col * chain_block;
        c1 = (c0 + chain_block < P.nlocal) ? c0 + chain_block : P.nlocal;
        lane = threadIdx.x & 63;
        wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        par = 0;
        failed = false;
        s_own = nullptr;
        const int j = lane & 15, g = lane >> 4, K = P.nmodel;
#pragma unroll
        for (int vb = 0; vb < 8; ++vb) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const size_t i = (size_t)t * 2048 + 1024 * h + 128 * wv + 16 * vb + j;
                // (the weights are the same for the four 16-lane rows of a wave: they live in
                // LDS, 16 KB, and leave 32 registers to the chain state of the owner waves)
                if (g == 0)
                    lds_ww[1024 * h + 128 * wv + 16 * vb + j] = P.w[i];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const int k = 4 * kb + g;
                    const double *src = (k < K) ? P.indep + (size_t)k * P.npad : P.data;
                    const double v = (k <= K) ? src[i] : 0.0;
                    if (kb < KR)
                        bop[vb][h][kb] = v;
                    else
                        lds_b[(2 * vb + h) * 512 + threadIdx.x] = v;
                }
            }
        }
        s_ww = lds_ww + 128 * wv + j;
    }

    // every wave of the workgroup, once per iteration; false (for all of them) = give up
    __device__ __forceinline__ bool tile_phase(const DevProblem &P, bool alive_in)
    {
        if (!alive_in || failed)
            *s_stop = 1;
        block_sync<8>();
        if (*s_stop != 0)
            return false;
        long long *ctl = const_cast<long long *>(P.ctl);
        const double *prop = P.rs_prop + (size_t)par * P.nlocal * RS_ROW;
        const int j = lane & 15, g = lane >> 4;
        const double *addr[KB];
        double aop[KB];
        auto request = [&](int cb) {
            const int lc = (cb + j < c1) ? cb + j : c1 - 1;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                addr[kb] = prop + (size_t)lc * RS_ROW + 4 * kb + g;
                aop[kb] = load_published(addr[kb]);
            }
        };
        // (the weights and the LDS part of the tile are read where they are used: an offset
        // the compiler cannot see through keeps it from hoisting the reads out of the loops
        // and parking the values in registers again)
        int wo = 0;
        // Step s = 0..15 of a block of chains: half h = s & 1 of block vb = s >> 1 of 16 lanes.
        // The product of a step, with the coefficients in a[]; bl = its operand from LDS
        auto product = [&](int st, const double(&a)[KB], double bl) {
            const int vb = st >> 1, h = st & 1;
            mc3_v4d m = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                m = __builtin_amdgcn_mfma_f64_16x16x4f64(
                    a[kb], (kb < KR) ? bop[vb][h][kb < KR ? kb : 0] : bl, m, 0, 0, 0);
            return m;
        };
        auto lds_operand = [&](int st) { return (KR < KB) ? s_b[wo + st * 512] : 0.0; };
        auto lds_weight = [&](int st) { return s_ww[wo + 1024 * (st & 1) + 16 * (st >> 1)]; };
        // sum over the 64 lanes of a group (lane 16 q + j): strides 32 and 16 were in-thread,
        // 8..1 run inside the 16-lane row; lane j = 0 has the defined order
        auto group_total = [&](const double(&acc)[2][4], int grp, int off) {
            double sr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double sg = acc[0][r] + acc[1][r];         // stride 16
                sg = add_ror<8>(sg);
                sg = add_ror<4>(sg);
                sg = add_ror<2>(sg);
                sr[r] = add_ror<1>(sg);
            }
            if (j == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s_w[(off + g + 4 * r) * 16 + 2 * wv + grp] = sr[r];
            }
        };
        // Software pipeline, written out.  FP64 matrix and vector instructions share one pipe
        // on this chip (tools/probes/mfma_f64_issue_probe.hip: their times add up), so what
        // can be hidden is latency, not work: the product of step s + 1 is issued before the
        // residuals of step s are squared, one matrix instruction then a few vector
        // instructions in turn; the LDS operands of step s + 2 are read during step s; the
        // cross-lane sum of the first group of 64 lanes runs between the matrix instructions
        // of the second.  The scheduling barriers keep that order, and with it two products'
        // worth of registers alive instead of sixteen.  (Carrying the pipeline across the
        // barrier at the end of a block of chains costs 25 more registers.)
        request(c0);
        int base = c0;                  // first chain of the window collected in s_w
        for (int cb = c0; cb < c1; cb += 16) {
            if (!wait_rows<KB>(addr, aop, ctl))
                failed = true;
            double a[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                a[kb] = aop[kb];
            if (cb + 16 < c1)
                request(cb + 16);              // the next block's rows travel behind this one
            asm volatile("; operands of this block" : "+v"(wo));
            double bl_next = lds_operand(1), w_cur = lds_weight(0), w_next = w_cur;
            mc3_v4d mc = product(0, a, lds_operand(0)), mn = mc;
            double accA[2][4], accB[2][4], tmp[4];     // [block of 16 lanes mod 2][chain g + 4 r]
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int q = (st >> 1) & 3, h = st & 1;
                double(&acc)[2][4] = (st < 8) ? accA : accB;
                __builtin_amdgcn_sched_barrier(0);
                const double bl = bl_next;
                if (st < 14)
                    bl_next = lds_operand(st + 2);
                if (st < 15) {
                    w_next = lds_weight(st + 1);
                    mn = product(st + 1, a, bl);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double res = mc[r] * w_cur;
                    // strides 32 (blocks 0 + 2, 1 + 3 of the group) as soon as both are there
                    if (q < 2) {
                        acc[q][r] = mc3_fma(res, res, h ? acc[q][r] : 0.0);
                    } else {
                        tmp[r] = mc3_fma(res, res, h ? tmp[r] : 0.0);
                        if (h == 1)
                            acc[q - 2][r] = acc[q - 2][r] + tmp[r];
                    }
                }
                if (st == 8)
                    group_total(accA, 0, cb - base);
                // (two matrix instructions first: the product these squares read was the last
                // thing issued in the previous step and needs a pipe pass to land)
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);             // MFMA
#pragma unroll
                for (int i = 2; i <= KB; ++i) {
                    if (st == 8)
                        __builtin_amdgcn_sched_group_barrier(0x002, 36, 0);    // VALU
                    else
                        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                    if (i < KB)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                mc = mn;
                w_cur = w_next;
            }
            group_total(accB, 1, cb - base);
            // The waves of the workgroup are NOT synchronised per block of chains: every wave
            // leaves its group totals in LDS and goes on (a barrier, an exchange and a
            // sixteen-lane sum per block cost ~10 % of the phase, and two waves that drift
            // apart share their SIMD's FP64 pipe better than two in step).  The totals are
            // added up -- 16 per chain, strides 8..1 -- once per window of RS_WCAP chains,
            // normally once per iteration.
            const int end = (cb + 16 < c1) ? cb + 16 : c1;
            if (end == c1 || end - base == RS_WCAP) {
                block_sync<8>();
                for (int ch0 = base + 4 * wv; ch0 < end; ch0 += 32) {
                    const int ch = ch0 + (lane >> 4);
                    double sg = s_w[((ch < end ? ch : end - 1) - base) * 16 + (lane & 15)];
                    sg = add_ror<8>(sg);
                    sg = add_ror<4>(sg);
                    sg = add_ror<2>(sg);
                    sg = add_ror<1>(sg);
                    if (mc3_bits(sg) == ROW_PENDING)   // (a NaN that looks like "not there yet")
                        sg = mc3_from_bits(0x7ff8000000000000ull);
                    if ((lane & 15) == 0 && ch < end)
                        __hip_atomic_store(&P.rs_part[(size_t)ch * P.rs_ntp + t], sg,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (end < c1)
                    block_sync<8>();            // (the next window overwrites these totals)
                base = end;
            }
        }
        par ^= 1;
        return true;
    }

};

// The same one-launch plan for polynomials of KX = 2..4 coefficients (tile_chisq_kernel's
// arithmetic: Horner on the vector unit, 4096 points per tile, lanes first -- one halving tree
// over the tile's 1024 strided partial sums).  Eight waves: thread l owns virtual lanes l and
// l + 512 with four points each (x, data, weight: 48 registers), so the first level of the
// tree is an addition inside the thread; 16 chains at a time, their 64 coefficients one per
// lane (v_readlane hands them to the scalar unit); the 512 partial sums of a chain cross the
// workgroup through LDS (two buffers of 16 x 512: one barrier per block of chains), wave v
// finishes chains 2 v and 2 v + 1.
template <int KX>
struct ResidentPoly : ResidentOwner {
    double x[2][4], d[2][4], w[2][4];
    double *s_x;            // LDS [2][16][512]
    int *s_stop;
    int t, c0, c1, wv;

    __device__ __forceinline__ void load(const DevProblem &P, int tile, int col, int chain_block,
                                         double *lds_x, int *lds_stop)
    {
        s_x = lds_x;
        s_stop = lds_stop;
        t = tile;
        c0 = col * chain_block;
        c1 = (c0 + chain_block < P.nlocal) ? c0 + chain_block : P.nlocal;
        lane = threadIdx.x & 63;
        wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        par = 0;
        failed = false;
        s_own = nullptr;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t i = (size_t)t * 4096 + threadIdx.x + 512 * v + 1024 * q;
                x[v][q] = P.indep[i];
                d[v][q] = P.data[i];
                w[v][q] = P.w[i];
            }
        }
    }

    __device__ __forceinline__ bool tile_phase(const DevProblem &P, bool alive_in)
    {
        if (!alive_in || failed)
            *s_stop = 1;
        block_sync<8>();
        if (*s_stop != 0)
            return false;
        long long *ctl = const_cast<long long *>(P.ctl);
        const double *prop = P.rs_prop + (size_t)par * P.nlocal * RS_ROW;
        const int l = threadIdx.x;
        const double *addr[1];
        double val[1];
        auto request = [&](int cb) {
            const int lc = (cb + (lane >> 2) < c1) ? cb + (lane >> 2) : c1 - 1;
            addr[0] = prop + (size_t)lc * RS_ROW + (lane & 3);
            val[0] = load_published(addr[0]);
        };
        request(c0);
        int buf = 0;
        for (int cb = c0; cb < c1; cb += 16, buf ^= 1) {
            if (!wait_rows<1>(addr, val, ctl))
                failed = true;
            const double cfv = val[0];
            if (cb + 16 < c1)
                request(cb + 16);              // the next block's rows travel behind this one
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                double cf[KX];
#pragma unroll
                for (int k = 0; k < KX; ++k)
                    cf[k] = read_lane(cfv, 4 * g + k);
                double acc[2];
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    acc[v] = 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        double m = cf[KX - 1];            // Horner from coefficient K-1 down
#pragma unroll
                        for (int k = KX - 2; k >= 0; --k)
                            m = mc3_fma(m, x[v][q], cf[k]);
                        const double r = (m - d[v][q]) * w[v][q];
                        acc[v] = mc3_fma(r, r, acc[v]);
                    }
                }
                s_x[(buf * 16 + g) * 512 + l] = acc[0] + acc[1];       // stride 512
            }
            block_sync<8>();
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int ch = 2 * wv + cc;
                double p[8];
#pragma unroll
                for (int w2 = 0; w2 < 8; ++w2)
                    p[w2] = s_x[(buf * 16 + ch) * 512 + lane + 64 * w2];
#pragma unroll
                for (int s2 = 4; s2 >= 1; s2 >>= 1) {                 // strides 256, 128, 64
#pragma unroll
                    for (int w2 = 0; w2 < s2; ++w2)
                        p[w2] = p[w2] + p[w2 + s2];
                }
                double tot = wave_sum(p[0]);                          // strides 32..1
                if (mc3_bits(tot) == ROW_PENDING)      // (a NaN that looks like "not there yet")
                    tot = mc3_from_bits(0x7ff8000000000000ull);
                if (lane == 0 && cb + ch < c1)
                    __hip_atomic_store(&P.rs_part[(size_t)(cb + ch) * P.rs_ntp + t], tot,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // (no second barrier: the next block writes the other buffer, and this one is not
            // written again before everybody has passed the next barrier)
        }
        par ^= 1;
        return true;
    }
};

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
template <int WPC, int MT, bool WL, bool MULTI, bool PERSIST, int SMP = -1, bool FAST = false,
          int PHASE = PH_FUSED, bool P2P = false, int TH = 0, bool FX = false, class H = NoHook>
__device__ __forceinline__ void run_chain(const DevProblem &P, const long long R0,
                                          const int nref, const int jb = 0, int je = -1,
                                          H *hook = nullptr, const int own_chain = 0,
                                          double *own_smem = nullptr)
{
    if (je < 0)
        je = P.thinning;
    static_assert(PHASE != PH_RESIDENT || (WPC == 1 && PERSIST && !FAST && !WL),
                  "the resident tiled plan runs the general one-wave persistent chain");
    static_assert(!FAST || (WPC == 1 && !WL && SMP >= 0 && MT >= 1 && MT <= 8),
                  "FAST needs one wave, Gaussian chisq, a compile-time sampler and degree");
    const int sampler = (SMP >= 0) ? SMP : P.sampler;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SmemLayout L = carve(PHASE == PH_RESIDENT ? own_smem : smem, P, 64 * WPC);
    MC3_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const bool wave0 = (WPC == 1) ? true : (tid < 64);
    const int lc = (PHASE == PH_RESIDENT) ? own_chain : blockIdx.x;   // local chain
    const int c = P.chain0 + lc;               // global chain id
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
    double xk = 0.0;
    if (mine)
        xk = (jb == 0) ? snap_first[(size_t)c * zs + lane] : P.X[(size_t)lc * nf + lane];
    double cur = P.chisq[lc];
    double best = P.best_chisq[lc];
    const long long nacc0 = P.numaccept[lc];
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
    if constexpr (FX)
        coefficient_map<NKF>(P, kf, fx);
    (void)kf; (void)fx;

    if constexpr (PERSIST && PHASE != PH_RESIDENT) {     // (resident: the kernel has looked)
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
            request_snooker(nxt, P.M0 + R0 * P.nchains);
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
    for (long long R = R0; R < R0 + nref && alive; ++R) {
    if constexpr (PERSIST) {
        if (P.test_abort >= 0 && R == P.test_abort && lc == 0) {     // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);             // (all lanes, same word)
            if constexpr (PHASE == PH_RESIDENT)
                give_up = true;                 // (leaves with its workgroup, below)
            else
                return;
        }
    }
    const double *snap = (R == 0) ? P.snap0
                                  : P.Z + (size_t)(P.M0 + (R - 1) * P.nchains) * zs;
    const long long zsize = P.M0 + R * P.nchains;
    const int jfirst = (TH == 1) ? 0 : jb;
    const int jlast = (TH == 1) ? 1 : je;
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
            // support-distribution draw, chain.py:185
            double normal_k = 0.0;
            if (need_normal && mine) {
                const Draw2 d = draw2(P.seed, iter, (uint32_t)c, (uint32_t)(3 + (lane >> 1)));
                const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(d.q0)));
                double sn, cs;
                mc3_det_sincos2pi(u01(d.q1), &sn, &cs);
                normal_k = (rad * ((lane & 1) ? sn : cs)) * pstep_k;
            }
            double jump_k = 0.0;
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
                    if constexpr (SNOOKER_AHEAD)
                        nxt = get_draws(batch, P, (long long)iter + 1, c);
                    else
                        nxt = make_draws(P, iter + 1, c, sampler);
                    alive = wait_rows<3, P2P>(addr, val, ctl_rw);
                    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see lean_demc_kernel
                    z1 = val[0];
                    z2 = val[1];
                    zk = val[2];
                    // the next iteration's rows: its history also holds this refresh's rows
                    // once this iteration is the last of the refresh
                    if (SNOOKER_AHEAD && alive)
                        request_snooker(nxt, (TH == 1 || j + 1 == jlast) ? zsize + P.nchains
                                                                         : zsize);
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
                if constexpr (PERSIST)
                    nxt = make_draws(P, iter + 1, c, sampler);
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
                    nxt = make_draws(P, iter + 1, c, sampler);
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
            if constexpr (PHASE == PH_RESIDENT) {
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
            if constexpr (PHASE == PH_RESIDENT) {
                // the tile phase of this iteration, with the whole workgroup; false for all of
                // its waves once one of them has given up
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
        if constexpr (PHASE == PH_RESIDENT)
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
                    nextchisq = nextchisq + prior_from_lanes<NKF>(P, npk, kf, fx);
            } else if constexpr (PHASE == PH_FINISH) {
                nextchisq = finish_chisq<MT>(P, lc);
            } else if constexpr (PHASE == PH_RESIDENT) {
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
                const double ratio = mc3_det_exp(0.5 * (cur - nextchisq)) * mrfactor;
                if (ratio > u) {                                     // chain.py:257-274
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
            if constexpr (PHASE == PH_RESIDENT) {
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
        if (mine) {
            if constexpr (P2P) {
                for (int pr = 0; pr < P.npeers; ++pr)               // publish everywhere
                    __hip_atomic_store(&P.peerZ[pr][(size_t)row * zs + lane], xk,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else if constexpr (PERSIST) {
                __hip_atomic_store(&P.Z[(size_t)row * zs + lane], xk, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);      // publish (sc1)
            } else {
                P.Z[(size_t)row * zs + lane] = xk;
            }
        }
        if (lane == 0) {
            if constexpr (P2P) {
                for (int pr = 0; pr < P.npeers; ++pr)
                    __hip_atomic_store(&P.peerLP[pr][row], -0.5 * cur, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                P.log_post[row] = -0.5 * cur;
            }
            P.zchain[row] = c;
        }
    }
    MC3_TICK(tk_prevend);
    }   // refreshes
    if (wave0) {
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
    if (PERSIST && threadIdx.x == 0 && blockIdx.x == 0 && (PHASE != PH_RESIDENT || lc == 0)) {
        P.stamps[8] += ta8; P.stamps[9] += ta9; P.stamps[10] += ta10;
        P.stamps[11] += ta11; P.stamps[12] += ta12; P.stamps[13] += ta13;
        P.stamps[14] += ta14; P.stamps[15] += ta15;
        if constexpr (PHASE == PH_RESIDENT) {
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

template <int WPC, int MT, bool WL, bool MULTI>
__global__ __launch_bounds__(64 * WPC) void step_kernel(const DevProblem P, const int roff)
{
    run_chain<WPC, MT, WL, MULTI, false>(P, P.ctl[0] + roff, 1);
    ring_out(P, roff);
}

// after the all-gather: the other ranks' rows of ring slot `roff` go to their place in Z
// (refresh ctl[0] + roff); own rows are already there
__global__ void ring_in_kernel(const DevProblem P, const int roff)
{
    const size_t zs = (size_t)P.zs;
    const double *src = P.xring + (size_t)roff * P.nchains * zs;
    double *dst = P.Z + (size_t)(P.M0 + (P.ctl[0] + roff) * P.nchains) * zs;
    const int nf = P.nfree;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.nchains * nf;
         i += gridDim.x * blockDim.x) {
        const int row = i / nf, k = i - row * nf;
        if (row < P.chain0 || row >= P.chain0 + P.nlocal)
            dst[(size_t)row * zs + k] = src[(size_t)row * zs + k];
    }
}

// log_post of this rank's chains over refreshes [R0, R0+nref) -> dense [nref][nlocal]
// (send buffer of ONE all-gather), and back from the gathered [nranks][nref][nlocal]
__global__ void logpost_pack_kernel(const DevProblem P, long long R0, long long nref,
                                    double *dense)
{
    const long long total = nref * P.nlocal;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / P.nlocal, lc = i - r * P.nlocal;
        dense[i] = P.log_post[P.M0 + (R0 + r) * P.nchains + P.chain0 + lc];
    }
}

__global__ void logpost_unpack_kernel(const DevProblem P, long long R0, long long nref,
                                      const double *gathered)
{
    const long long total = nref * P.nchains;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / P.nchains, ch = i - r * P.nchains;
        const long long rank = ch / P.nlocal, lc = ch - rank * P.nlocal;
        if (rank != P.chain0 / P.nlocal)
            P.log_post[P.M0 + (R0 + r) * P.nchains + ch] =
                gathered[(rank * nref + r) * P.nlocal + lc];
    }
}

// many Z refreshes per launch with in-launch hand-off of the new rows; every
// workgroup of the grid must be resident (the host checks before choosing this path)
template <int WPC, int MT, bool WL, bool MULTI, bool P2P = false>
__global__ __launch_bounds__(64 * WPC) void persist_kernel(const DevProblem P, const int nref)
{
    run_chain<WPC, MT, WL, MULTI, true, -1, false, PH_FUSED, P2P>(P, P.ctl[0], nref);
}


// The tiled plan in one launch (ResidentTile above): grid = tiles x columns of chains, eight
// waves; wave v of workgroup w also owns chain w + v * workgroups.  P2P: the ensemble is
// sharded over GPUs, each running this kernel over its own chains; owners publish their Z
// rows into every rank's replica (run_chain P2P), proposals and tile sums stay local.  LDS:
// group totals
// [RS_WCAP][16], the give-up word, a run_chain area per wave, the tile's 2048 weights, and
// (three blocks of operand rows) the last block, [16][512].
template <int KB, bool P2P = false>
__global__ __launch_bounds__(512) void resident_tile_kernel(const DevProblem P, const int nref,
                                                            const int chain_block)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int OWN = 64 + MC3HIP_MAX_PARAMS + 2 + 48 + 4;     // carve() of a one-wave chain
    constexpr int W = RS_WCAP * 16;                              // group totals
    int *s_stop = reinterpret_cast<int *>(smem + W);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    ResidentTile<KB> T;
    T.load(P, blockIdx.x, blockIdx.y, chain_block, smem, s_stop, smem + W + 2 + 8 * OWN,
           smem + W + 2 + 8 * OWN + 2048);
    // a launch queued behind one that gave up must not run: its inputs are missing
    if (threadIdx.x == 0)
        *s_stop = (__hip_atomic_load(&P.ctl[CTL_ABORT], __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1 : 0;
    block_sync<8>();
    if (*s_stop != 0)
        return;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    const int own_chain = wg + (int)(gridDim.x * gridDim.y) * wv;
    if (own_chain < P.nlocal) {
        T.set_owner(P, own_chain, smem + W + 2 + wv * OWN);
        run_chain<1, MT_LINEAR, false, false, true, -1, false, PH_RESIDENT, P2P, 0, false,
                  ResidentTile<KB>>(P, P.ctl[0], nref, 0, -1, &T, own_chain,
                                    smem + W + 2 + wv * OWN);
    } else {
        // no chain of its own: the tile phase of every iteration, in step with the owners
        for (int R = 0; R < nref; ++R)
            for (int j = 0; j < P.thinning; ++j)
                if (!T.tile_phase(P, true))
                    return;
    }
}

template <int KX, bool P2P = false>
__global__ __launch_bounds__(512) void resident_poly_kernel(const DevProblem P, const int nref,
                                                            const int chain_block)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int OWN = 64 + MC3HIP_MAX_PARAMS + 2 + 48 + 4;     // carve() of a one-wave chain
    constexpr int X = 2 * 16 * 512;                              // the exchange buffers
    int *s_stop = reinterpret_cast<int *>(smem + X);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    ResidentPoly<KX> T;
    T.load(P, blockIdx.x, blockIdx.y, chain_block, smem, s_stop);
    // a launch queued behind one that gave up must not run: its inputs are missing
    if (threadIdx.x == 0)
        *s_stop = (__hip_atomic_load(&P.ctl[CTL_ABORT], __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1 : 0;
    block_sync<8>();
    if (*s_stop != 0)
        return;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    const int own_chain = wg + (int)(gridDim.x * gridDim.y) * wv;
    if (own_chain < P.nlocal) {
        T.set_owner(P, own_chain, smem + X + 2 + wv * OWN);
        run_chain<1, KX, false, false, true, -1, false, PH_RESIDENT, P2P, 0, false,
                  ResidentPoly<KX>>(P, P.ctl[0], nref, 0, -1, &T, own_chain,
                                    smem + X + 2 + wv * OWN);
    } else {
        for (int R = 0; R < nref; ++R)
            for (int j = 0; j < P.thinning; ++j)
                if (!T.tile_phase(P, true))
                    return;
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

// the lean instance (see run_chain): one wave per chain, MT free polynomial
// coefficients, Gaussian chisq, ndata <= 1024, no priors/shared/fixed/epsilon/trace
// (MULTI: more than 1024 points; the first tile stays in registers, the rest is streamed)
template <int MT, int SMP, bool P2P = false, int TH = 0, bool FX = false, bool MULTI = false>
__global__ __launch_bounds__(64) void persist_fast_kernel(const DevProblem P, const int nref)
{
    run_chain<1, MT, false, MULTI, true, SMP, true, PH_FUSED, P2P, TH, FX>(P, P.ctl[0], nref);
}

// chisq for arbitrary parameter vectors (Chain.eval_model(ret='chisq'))
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
template <int MT, bool P2P, bool FX = false, bool MULTI = false>
__global__ __launch_bounds__(64) void lean_demc_kernel(const DevProblem P, const int nref)
{
    const int nf = FX ? P.nfree : MT;        // FX: some coefficients are fixed (nfree < MT)
    constexpr int ZS = ZROW;                 // doubles per Z row: one 128-byte line (NF <= 16)
    constexpr int LM = P2P ? 1 : 0;
    const int lane = threadIdx.x;
    const int lc = blockIdx.x;
    const int c = P.chain0 + lc;
    const bool mine = lane < nf;
    const int kk = mine ? lane : 0;
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(
            &ctl_rw[CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
        return;
    MC3_STAMP(0);
    DataTile<1, MT> tile;
    tile.load(P);
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
    int kf[MT];
    double fx[MT];
    if constexpr (FX)
        coefficient_map<MT>(P, kf, fx);
    (void)kf; (void)fx;
    // log_post / zchain of the last (up to) 64 refreshes wait in lane (R - R0) mod 64 and
    // are written 64 at a time: nobody reads them inside the launch
    double lpbuf = 0.0;
    MC3_STAMP(1);

    // draws of the first refresh
    Draw2 d = draw2(P.seed, (uint64_t)R0, (uint32_t)c, 0u);
    double u = u01(d.q1);
    int r1, r2;
    demc_partners(d.q0, n, c, r1, r2);
    int off1 = r1 * ZS + kk, off2 = r2 * ZS + kk;

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
                for (int pr = 0; pr < P.npeers; ++pr)
                    __hip_atomic_store(&P.peerLP[pr][roff], lpbuf, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                lp_row[(size_t)lane * n] = lpbuf;
            }
            zc_row[(size_t)lane * n] = c;
        }
    };
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
        __builtin_amdgcn_s_sleep(1);
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
        const double u_now = u;
        __builtin_amdgcn_sched_barrier(0);
        // ... and behind them everything of the next refresh that needs no chain state
        d = draw2(P.seed, (uint64_t)(R + 1), (uint32_t)c, 0u);
        pin_scalar(d.q0);
        pin_scalar(d.q1);
        u = u01(d.q1);                                               // chain.py:259
        demc_partners(d.q0, n, c, r1, r2);
        off1 = r1 * ZS + kk;
        off2 = r2 * ZS + kk;
        asm volatile("; pin" : "+v"(u), "+v"(off1), "+v"(off2));
        __builtin_amdgcn_sched_barrier(0);
#ifdef MC3HIP_STAMPS
        alive = poll_ring<KR, LM>(addr, ring, val, ctl_rw, &ta14);
#else
        alive = poll_ring<KR, LM>(addr, ring, val, ctl_rw);
#endif
        MC3_TICK(tk1);
        if (__builtin_expect(!alive, 0))
            break;
        // proposal and boundaries, chain.py:232-243
        const double jump_k = gamma * (val[0] - val[1]) + feps * 0.0;
        const double npk = xk + jump_k;
        const bool outk = mine && (npk < pmin_k || npk > pmax_k);
        if (__builtin_expect(!__any(outk), 1)) {
            Model<MT> model;
            if constexpr (!FX) {
                model.init_from_lanes(P, npk);                 // every coefficient is free
            } else {
                model.init_from_lanes(P, npk, kf, fx);         // FX: some are fixed
            }
            double nextchisq =
                block_gauss_model<1, MT, MULTI, MULTI>(P, model, nullptr, nullptr, tile,
                                                       &tile1);          // chain.py:249
            if constexpr (FX)
                nextchisq = nextchisq + prior_from_lanes<MT>(P, npk, kf, fx);
            MC3_TICK(tk2);
            const double ratio = mc3_det_exp(0.5 * (cur - nextchisq)) * 1.0;
            const bool acc = ratio > u_now;                          // chain.py:257-274
            xk = acc ? npk : xk;
            if (__builtin_expect(__any(acc && nextchisq < best), 0)) {
                best = nextchisq;
                if (mine)
                    P.bestx[(size_t)lc * nf + lane] = xk;
                if (lane == 0)
                    P.best_iter[lc] = R;
            }
            cur = acc ? nextchisq : cur;
            nacc += acc ? 1 : 0;
        } else if (outk) {
            atomicAdd(&P.outbounds[lane], 1ull);                     // chain.py:242-243
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
        __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0), already true
        // Z append, chain.py:276-289: publish the row ...
        if constexpr (P2P) {
            const size_t zoff = (size_t)(zblock - P.Z) + (size_t)c * ZS + lane;
            if (mine)
                for (int pr = 0; pr < P.npeers; ++pr)                // ... into every replica
                    __hip_atomic_store(&P.peerZ[pr][zoff], xk, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
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

template <int WPC, int MT, bool WL, bool LARGE = false>
__global__ __launch_bounds__(64 * WPC) void eval_chisq_kernel(const DevProblem P,
                                                              const double *params,
                                                              double *out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SmemLayout L = carve(smem, P, 64 * WPC);
    DataTile<WPC, MT> tile;
    if constexpr (!WL)
        tile.load(P);
    if constexpr (WL)
        wavelet_stage<WPC, MT>(P, L.xs);         // (the __syncthreads below covers it)
    for (int q = threadIdx.x; q < P.nparams; q += 64 * WPC)
        L.nextp[q] = params[(size_t)blockIdx.x * P.nparams + q];
    __syncthreads();
    const double c = block_chisq<WPC, MT, WL, !WL || LARGE>(P, L.nextp, L.part, L.flag, L.tab,
                                                            L.halo, L.xs, tile);
    if (threadIdx.x == 0)
        out[blockIdx.x] = c;
}

template <int MT>
__global__ void eval_model_kernel(const DevProblem P, const double *params, double *out)
{
    __shared__ double s_p[MC3HIP_MAX_PARAMS];
    for (int q = threadIdx.x; q < P.nparams; q += blockDim.x)
        s_p[q] = params[q];
    __syncthreads();
    Model<MT> model;
    model.init(P, s_p);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.ndata;
         i += gridDim.x * blockDim.x)
        out[i] = model.at(i);
}

// standalone D4 transform (ms.dwt_daub4): one workgroup, global-memory ping-pong
__global__ void daub4_kernel(double *a, double *tmp, int n, int inverse)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    if (!inverse) {
        for (int nn = n; nn >= 4; nn >>= 1) {
            const int nh = nn >> 1;
            for (int i = tid; i < nh; i += nt) {
                double s, d;
                d4_pair(a[2 * i], a[2 * i + 1], a[(2 * i + 2) & (nn - 1)],
                        a[(2 * i + 3) & (nn - 1)], s, d);
                tmp[i] = s;
                tmp[i + nh] = d;
            }
            __syncthreads();
            for (int i = tid; i < nn; i += nt)
                a[i] = tmp[i];
            __syncthreads();
        }
    } else {
        for (int nn = 4; nn <= n; nn <<= 1) {                        // wavelet.h:37-43
            const int nh = nn >> 1;
            for (int i = tid; i < nh; i += nt) {
                const int im = (i + nh - 1) & (nh - 1);
                const double sp = a[im], dp = a[im + nh], sc = a[i], dc = a[i + nh];
                double e = D4C2 * sp;
                e = mc3_fma(D4C1, dp, e);
                e = mc3_fma(D4C0, sc, e);
                e = mc3_fma(D4C3, dc, e);
                double o = D4C3 * sp;
                o = mc3_fma(-D4C0, dp, o);
                o = mc3_fma(D4C1, sc, o);
                o = mc3_fma(-D4C2, dc, o);
                tmp[2 * i] = e;
                tmp[2 * i + 1] = o;
            }
            __syncthreads();
            for (int i = tid; i < nn; i += nt)
                a[i] = tmp[i];
            __syncthreads();
        }
    }
}

// ---- standalone likelihood kernels: the public functions of the reference's
// _chisq / _dwt extension modules on arbitrary model arrays --------------------
// residuals, src_c/_chisq.c:37-79: (model-data)/uncert, then the prior residuals
__global__ void residuals_kernel(const double *model, const double *data, const double *uncert,
                                 int n, const double *off, const double *plow,
                                 const double *pup, int np, double *out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n + np; i += gridDim.x * blockDim.x) {
        if (i < n) {
            out[i] = (model[i] - data[i]) / uncert[i];
        } else {
            const int j = i - n;
            out[i] = (off[j] > 0.0) ? off[j] / pup[j] : off[j] / plow[j];
        }
    }
}

// Sum of squares of v[0..n) with optional per-element weights given by scale
// (wavelet: weight = inv[floor(log2 j)] for j >= 2, 0 below): one workgroup of 1024
// threads, canonical order (strided partials over 1024 lanes, halving tree).
__global__ __launch_bounds__(1024) void sumsq_kernel(const double *v, int n, const double *inv,
                                                     double *out)
{
    __shared__ double s_part[1024];
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += 1024) {
        if (inv == nullptr) {
            acc[0] = mc3_fma(v[i], v[i], acc[0]);
        } else if (i >= 2) {
            acc[0] = mc3_fma(v[i] * v[i], inv[31 - __clz(i)], acc[0]);
        }
    }
    const double t = lane_tree<16>(acc, s_part);
    if (threadIdx.x == 0)
        out[0] = t;
}

__global__ void diff_kernel(const double *a, const double *b, int n, int npad, double *out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += gridDim.x * blockDim.x)
        out[i] = (i < n) ? a[i] - b[i] : 0.0;
}

// Per-chain mean and (population) variance of every free parameter over the rows a
// chain has written, skipping its first `burn` rows: the reductions the
// Gelman-Rubin test needs (mc3/stats/gelman.py:12-92) without copying Z to the host.
// One workgroup per chain; two passes (mean, then squared deviations).
__global__ __launch_bounds__(256) void chain_moments_kernel(const double *Z, long long M0,
                                                            int nchains, int nf, int zs,
                                                            long long nref, long long burn,
                                                            double *mean, double *var)
{
    __shared__ double s_red[256];
    const int c = blockIdx.x, t = threadIdx.x;
    const long long n = nref - burn;
    for (int k = 0; k < nf; ++k) {
        double acc = 0.0;
        for (long long r = burn + t; r < nref; r += 256)
            acc += Z[(size_t)(M0 + r * nchains + c) * zs + k];
        s_red[t] = acc;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if (t < s2)
                s_red[t] += s_red[t + s2];
            __syncthreads();
        }
        const double mu = s_red[0] / (double)n;
        __syncthreads();
        acc = 0.0;
        for (long long r = burn + t; r < nref; r += 256) {
            const double d = Z[(size_t)(M0 + r * nchains + c) * zs + k] - mu;
            acc += d * d;
        }
        s_red[t] = acc;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if (t < s2)
                s_red[t] += s_red[t + s2];
            __syncthreads();
        }
        if (t == 0) {
            mean[(size_t)c * nf + k] = mu;
            var[(size_t)c * nf + k] = s_red[0] / (double)n;
        }
        __syncthreads();
    }
}

// ---- posterior summary on the device (mc3/stats/stats.py:805-852,876-964) -----------------
// What update_output / calc_sample_statistics compute from the burned posterior with
// NumPy -- per-parameter mean, standard deviation, extrema, median and central percentiles
// -- without copying Z anywhere: two deterministic reduction passes and an exact
// multi-rank radix select over sortable 64-bit keys.
__device__ __forceinline__ uint64_t sortable_key(double v)
{
    const uint64_t b = mc3_bits(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);     // total order of finite doubles
}
__device__ __forceinline__ double key_to_double(uint64_t k)
{
    return mc3_from_bits((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k);
}

// pass 1: per block and parameter, sum / min / max of rows [row0, row0 + n); also the
// sortable keys, dense per parameter (keys[k][i]), for the selection passes
__global__ __launch_bounds__(256) void summary_pass1_kernel(const double *Z, long long row0,
                                                            long long n, int nf, int zs,
                                                            uint64_t *keys, double *part)
{
    __shared__ double s_sum[256], s_min[256], s_max[256];
    const int t = threadIdx.x;
    for (int k = 0; k < nf; ++k) {
        double sum = 0.0, lo = INFINITY, hi = -INFINITY;
        for (long long i = (long long)blockIdx.x * 256 + t; i < n; i += (long long)gridDim.x * 256) {
            const double v = Z[(size_t)(row0 + i) * zs + k];
            keys[(size_t)k * n + i] = sortable_key(v);
            sum += v;
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
        s_sum[t] = sum; s_min[t] = lo; s_max[t] = hi;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if (t < s2) {
                s_sum[t] += s_sum[t + s2];
                s_min[t] = s_min[t + s2] < s_min[t] ? s_min[t + s2] : s_min[t];
                s_max[t] = s_max[t + s2] > s_max[t] ? s_max[t + s2] : s_max[t];
            }
            __syncthreads();
        }
        if (t == 0) {
            double *o = part + ((size_t)k * gridDim.x + blockIdx.x) * 4;
            o[0] = s_sum[0]; o[1] = s_min[0]; o[2] = s_max[0];
        }
        __syncthreads();
    }
}

// between the passes: block partials -> mean / min / max per parameter (one block)
__global__ __launch_bounds__(256) void summary_finish1_kernel(const double *part, int nblocks,
                                                              int nf, long long n, double *res)
{
    __shared__ double s_sum[256], s_min[256], s_max[256];
    const int t = threadIdx.x;
    for (int k = 0; k < nf; ++k) {
        double sum = 0.0, lo = INFINITY, hi = -INFINITY;
        for (int b = t; b < nblocks; b += 256) {
            const double *o = part + ((size_t)k * nblocks + b) * 4;
            sum += o[0];
            lo = o[1] < lo ? o[1] : lo;
            hi = o[2] > hi ? o[2] : hi;
        }
        s_sum[t] = sum; s_min[t] = lo; s_max[t] = hi;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if (t < s2) {
                s_sum[t] += s_sum[t + s2];
                s_min[t] = s_min[t + s2] < s_min[t] ? s_min[t + s2] : s_min[t];
                s_max[t] = s_max[t + s2] > s_max[t] ? s_max[t + s2] : s_max[t];
            }
            __syncthreads();
        }
        if (t == 0) {
            res[4 * k] = s_sum[0] / (double)n;          // mean
            res[4 * k + 2] = s_min[0];
            res[4 * k + 3] = s_max[0];
        }
        __syncthreads();
    }
}

// pass 2: sum of squared deviations from the mean (np.var's second pass)
__global__ __launch_bounds__(256) void summary_pass2_kernel(const double *Z, long long row0,
                                                            long long n, int nf, int zs,
                                                            const double *res, double *part)
{
    __shared__ double s_sum[256];
    const int t = threadIdx.x;
    for (int k = 0; k < nf; ++k) {
        const double mu = res[4 * k];
        double sum = 0.0;
        for (long long i = (long long)blockIdx.x * 256 + t; i < n; i += (long long)gridDim.x * 256) {
            const double d = Z[(size_t)(row0 + i) * zs + k] - mu;
            sum += d * d;
        }
        s_sum[t] = sum;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if (t < s2)
                s_sum[t] += s_sum[t + s2];
            __syncthreads();
        }
        if (t == 0)
            part[((size_t)k * gridDim.x + blockIdx.x) * 4 + 3] = s_sum[0];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void summary_finish2_kernel(const double *part, int nblocks,
                                                              int nf, long long n, double *res)
{
    __shared__ double s_sum[256];
    const int t = threadIdx.x;
    for (int k = 0; k < nf; ++k) {
        double sum = 0.0;
        for (int b = t; b < nblocks; b += 256)
            sum += part[((size_t)k * nblocks + b) * 4 + 3];
        s_sum[t] = sum;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if (t < s2)
                s_sum[t] += s_sum[t + s2];
            __syncthreads();
        }
        if (t == 0)
            res[4 * k + 1] = s_sum[0] / (double)n;      // population variance
        __syncthreads();
    }
}

// Exact order statistics by radix selection, 8 bits per pass from the top: every
// (parameter, rank) pair carries the key prefix fixed so far and the rank left inside it;
// a pass histograms the next byte of the keys that share a pair's prefix (LDS, then one
// atomic flush per block), a one-block kernel picks each pair's byte.  MAXR ranks at most.
constexpr int SEL_MAXR = 8;
struct SelectState {
    uint64_t prefix[SEL_MAXR];      // key bits fixed so far (high part), rest zero
    long long rank[SEL_MAXR];       // rank inside the keys that share the prefix
};

__global__ __launch_bounds__(256) void select_hist_kernel(const uint64_t *keys, long long n, int nr,
                                                          int pass, const SelectState *st,
                                                          unsigned int *hist)
{
    __shared__ unsigned int s_h[SEL_MAXR * 256];
    const int k = blockIdx.y, t = threadIdx.x;
    for (int q = t; q < nr * 256; q += 256)
        s_h[q] = 0;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const uint64_t himask = pass ? ~0ull << (shift + 8) : 0ull;
    uint64_t pre[SEL_MAXR];
    bool first[SEL_MAXR];           // pairs that share a prefix share the count: the first adds
    for (int r = 0; r < nr; ++r) {
        pre[r] = st[k].prefix[r] & himask;
        first[r] = true;
        for (int r2 = 0; r2 < r; ++r2)
            first[r] = first[r] && pre[r2] != pre[r];
    }
    const uint64_t *col = keys + (size_t)k * n;
    for (long long i = (long long)blockIdx.x * 256 + t; i < n; i += (long long)gridDim.x * 256) {
        const uint64_t key = col[i];
        const unsigned int digit = (unsigned int)(key >> shift) & 255u;
        for (int r = 0; r < nr; ++r)
            if (first[r] && (key & himask) == pre[r])
                atomicAdd(&s_h[r * 256 + digit], 1u);
    }
    __syncthreads();
    for (int q = t; q < nr * 256; q += 256)
        if (s_h[q])
            atomicAdd(&hist[(size_t)k * SEL_MAXR * 256 + q], s_h[q]);
}

__global__ __launch_bounds__(256) void select_pick_kernel(int nr, int pass, SelectState *st,
                                                          unsigned int *hist)
{
    const int k = blockIdx.x, t = threadIdx.x;
    const int shift = 56 - 8 * pass;
    const uint64_t himask = pass ? ~0ull << (shift + 8) : 0ull;
    __shared__ unsigned int s_h[SEL_MAXR * 256];
    unsigned int *h = hist + (size_t)k * SEL_MAXR * 256;
    // a pair without counts of its own reads those of the first pair with its prefix
    for (int q = t; q < nr * 256; q += 256) {
        const int r = q >> 8;
        int owner = r;
        for (int r2 = r - 1; r2 >= 0; --r2)
            if ((st[k].prefix[r2] & himask) == (st[k].prefix[r] & himask))
                owner = r2;
        s_h[q] = h[owner * 256 + (q & 255)];
    }
    __syncthreads();
    if (t < nr) {
        long long left = st[k].rank[t];
        unsigned int d = 0;
        for (; d < 255; ++d) {
            const unsigned int cnt = s_h[t * 256 + d];
            if (left < (long long)cnt)
                break;
            left -= cnt;
        }
        st[k].rank[t] = left;
        st[k].prefix[t] = (st[k].prefix[t] & himask) | ((uint64_t)d << shift);
    }
    __syncthreads();
    for (int q = t; q < SEL_MAXR * 256; q += 256)
        h[q] = 0;
}

__global__ void select_result_kernel(const SelectState *st, int nf, int nr, double *out)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nf * nr)
        out[q] = key_to_double(st[q / nr].prefix[q % nr]);      // [nf][nr]
}

// chisq = -2 (log_post - log_prior) of rows [row0, row0 + n) (mc3/stats/stats.py:821 with
// log_prior of :287-392: Gaussian priors with two-sided widths where priorlow, priorup > 0;
// the reference's log-uniform branch where priorlow < 0), and the packed dense copy of the
// rows' free parameters -- everything update_output needs per row, in one pass
__global__ void rows_chisq_kernel(const DevProblem P, long long row0, long long n, double *chisq)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const double *z = P.Z + (size_t)(row0 + i) * P.zs;
        double s = 0.0;
        for (int k = 0; k < P.nfree; ++k) {
            const int j = P.ifree[k];
            double d = 0.0;
            if (P.priorlow[j] > 0.0 && P.priorup[j] > 0.0) {
                const double off = z[k] - P.prior[j];
                d = (off == 0.0) ? 0.0 : off / (off < 0.0 ? P.priorlow[j] : P.priorup[j]);
            } else if (P.priorlow[j] < 0.0) {
                d = 2.0 * mc3_det_log(z[k]);
            }
            s = s + d * d;
        }
        chisq[i] = -2.0 * (P.log_post[row0 + i] - (-0.5 * s));
    }
}

// Gaussian kernel density of a 1D sample at m points (scipy.stats.gaussian_kde.evaluate,
// called by cred_region, mc3/stats/stats.py:395-467): pdf[j] = norm * sum_i exp(-(x_j -
// s_i)^2 / (2 cov)).  One block per evaluation point.
__global__ __launch_bounds__(256) void kde_kernel(const double *sample, int n, const double *x,
                                                  double inv2cov, double norm, double *pdf)
{
    __shared__ double s_sum[256];
    const int t = threadIdx.x;
    const double xj = x[blockIdx.x];
    double sum = 0.0;
    for (int i = t; i < n; i += 256) {
        const double d = xj - sample[i];
        sum += mc3_det_exp(-(d * d) * inv2cov);
    }
    s_sum[t] = sum;
    __syncthreads();
    for (int s2 = 128; s2 >= 1; s2 >>= 1) {
        if (t < s2)
            s_sum[t] += s_sum[t + s2];
        __syncthreads();
    }
    if (t == 0)
        pdf[blockIdx.x] = s_sum[0] * norm;
}

// dense [rows][nf] (what the C-ABI exchanges) <-> padded [rows][zs] (device layout of Z)
__global__ void pack_rows_kernel(double *dense, const double *padded, long long nrows, int nf,
                                 int zs)
{
    const long long total = nrows * nf;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
        dense[i] = padded[(i / nf) * zs + (i % nf)];
}

__global__ void unpack_rows_kernel(double *padded, const double *dense, long long nrows, int nf,
                                   int zs)
{
    const long long total = nrows * nf;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
        padded[(i / nf) * zs + (i % nf)] = dense[i];
}

// ---- initial population on the device (mc3/mcmc_driver.py:229-270) -------------------
// Trial t of the kick-off: nfree draws around the initial guess ('normal': params +
// pstep * N(0,1), Box-Muller pairs; 'uniform': between the boundaries), keyed like every
// other draw by Philox(seed; counter = (t, chain = 0xffffffff, slot)) -- a chain id no walker
// has.  One thread per trial writes the full parameter vector (fixed values kept, shared
// parameters copied, chain.py:246-247) for the chisq kernel, the free parameters, and
// whether the FREE parameters are inside their boundaries.
constexpr uint32_t KICKOFF_STREAM = 0xffffffffu;

__global__ void init_trials_kernel(const DevProblem P, int kickoff_uniform, uint64_t seed,
                                   long long t0, int ntrial, double *full, double *freep,
                                   int32_t *inside)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntrial)
        return;
    const uint64_t t = (uint64_t)(t0 + i);
    double *row = full + (size_t)i * P.nparams;
    for (int j = 0; j < P.nparams; ++j)
        row[j] = P.params[j];
    int ok = 1;
    for (int k2 = 0; 2 * k2 < P.nfree; ++k2) {
        const Draw2 d = draw2(seed, t, KICKOFF_STREAM, (uint32_t)k2);
        double a, b;
        if (kickoff_uniform) {
            a = u01(d.q0);
            b = u01(d.q1);
        } else {
            const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(d.q0)));
            double sn, cs;
            mc3_det_sincos2pi(u01(d.q1), &sn, &cs);
            a = rad * cs;
            b = rad * sn;
        }
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * k2 + h;
            if (k >= P.nfree)
                break;
            const double r = h ? b : a;
            const int j = P.ifree[k];
            const double v = kickoff_uniform ? P.fmin[k] + (P.fmax[k] - P.fmin[k]) * r
                                             : P.params[j] + P.fstep[k] * r;
            row[j] = v;
            freep[(size_t)i * P.nfree + k] = v;
            ok &= !(v < P.fmin[k] || v > P.fmax[k]);
        }
    }
    for (int sidx = 0; sidx < P.nshare; ++sidx)
        row[P.ishare[sidx]] = row[P.share_src[sidx]];
    inside[i] = ok;
}

// The reference's acceptance rule, in trial order (one thread: the rule is sequential and
// the work is a few operations per trial).  A trial counts as a failure when a parameter
// lies outside its boundaries -- tested on the full vector BEFORE the shared parameters are
// refreshed, i.e. with the copies the last in-bounds trial left (mcmc_driver.py:249-255) --
// or when its chisq is not finite; the loop ends with M0 rows or after 100 M0 failures.
// st[0] = rows filled, st[1] = failures, st[2] = trials consumed; stale[] = the shared
// parameters' current copies.
__global__ void init_scan_kernel(const DevProblem P, int ntrial, const double *full,
                                 const double *freep, const int32_t *inside,
                                 const double *chisq, long long *st, double *stale,
                                 double *Z0, double *lp0)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
        return;
    long long filled = st[0], failed = st[1], used = st[2];
    const long long M0 = P.M0, nmax = 100 * P.M0;
    for (int i = 0; i < ntrial; ++i) {
        if (filled == M0 || failed == nmax)
            break;
        used += 1;
        bool ok = inside[i] != 0;
        for (int sidx = 0; sidx < P.nshare && ok; ++sidx) {
            const int j = P.ishare[sidx];
            ok = !(stale[sidx] > P.pmax[j] || stale[sidx] < P.pmin[j]);
        }
        if (!ok) {
            failed += 1;
            continue;
        }
        for (int sidx = 0; sidx < P.nshare; ++sidx)
            stale[sidx] = full[(size_t)i * P.nparams + P.share_src[sidx]];
        const double lp = -0.5 * chisq[i];
        if ((mc3_bits(lp) & 0x7ff0000000000000ull) == 0x7ff0000000000000ull) {
            failed += 1;                                  // inf or NaN
            continue;
        }
        for (int k = 0; k < P.nfree; ++k)
            Z0[(size_t)filled * P.nfree + k] = freep[(size_t)i * P.nfree + k];
        lp0[filled] = lp;
        filled += 1;
    }
    st[0] = filled;
    st[1] = failed;
    st[2] = used;
}

// per-chain state to / from the checkpoint area (8-byte words; the seven items back to back
// in `ckpt`): one launch instead of seven asynchronous device-to-device copies per run
struct CheckpointItems {
    unsigned long long *ptr[7];
    unsigned long long words[7];
};

__global__ void checkpoint_kernel(CheckpointItems it, unsigned long long *ckpt, int save)
{
    unsigned long long at = 0;
    for (int k = 0; k < 7; ++k) {
        for (unsigned long long i = blockIdx.x * blockDim.x + threadIdx.x; i < it.words[k];
             i += (unsigned long long)gridDim.x * blockDim.x) {
            if (save)
                ckpt[at + i] = it.ptr[k][i];
            else
                it.ptr[k][i] = ckpt[at + i];
        }
        at += it.words[k];
    }
}

__global__ void set_ctl_kernel(long long *ctl, long long base, long long run0)
{
    ctl[0] = base;
    if (run0 >= 0)
        ctl[1] = run0;      // ctl[CTL_ABORT] is only ever cleared by the host (settle)
}

__global__ void selftest_math_kernel(int op, const double *in, int n, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double x = in[i];
    double r, s, c;
    switch (op) {
    case 0: r = mc3_det_exp(x); break;
    case 1: r = mc3_det_log(x); break;
    case 2: r = mc3_det_exp2(x); break;
    case 3: mc3_det_sincos2pi(x, &s, &c); r = s; break;
    case 4: mc3_det_sincos2pi(x, &s, &c); r = c; break;
    case 5: r = __builtin_sqrt(x); break;
    default: r = 1.0 / x; break;
    }
    out[i] = r;
}

__global__ void selftest_philox_kernel(const uint32_t *ctr, int n, uint32_t k0, uint32_t k1,
                                       uint32_t *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    uint32_t w[4];
    philox4x32_10(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], k0, k1, w);
    for (int q = 0; q < 4; ++q)
        out[4 * i + q] = w[q];
}

}  // namespace mc3hip
