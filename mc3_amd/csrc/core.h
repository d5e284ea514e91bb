// core.h -- what every chain-update kernel shares (gfx950, wave64): the device view of a
// problem, model evaluation, the canonical summation tree, the register-resident data tile.
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
// Diagnostic instrumentation (phase stamps, the timeline of a round, the pipelined wavelet
// kernel's slot accounting) and the tuning constants of resident_async_kernel exist in a
// DEVELOPMENT build only (-DMC3HIP_DEV: tools/stamps.py, tools/pipe_diag.py, tools/variant_lib.py,
// python -m mc3_amd.build --dev): without that flag none of these macros has any effect, whatever
// else is on the command line.
#ifndef MC3HIP_DEV
#undef MC3HIP_STAMPS
#undef MC3HIP_TICKS_ONLY
#undef MC3HIP_NO_LSTAMP
#undef MC3HIP_TIMELINE
#undef MC3HIP_PIPE_DIAG
#undef MC3HIP_RA_TAKE_AT
#undef MC3HIP_RA_TAKE_AGAIN
#undef MC3HIP_RA_PRIO
#undef MC3HIP_CW_PRIO
#undef MC3HIP_TAIL_PRIO
#endif
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
// MT_LINK0 + K: the same with K (1..16) columns known at compile time and the design matrix's
// first 1024-point tile resident in registers -- the specialisations made at run time
// (mc3_amd/jit.py) for the hand-laid-out one-wave loops
constexpr int MT_LINK0 = 200;
constexpr bool mt_is_link(int MT) { return MT > MT_LINK0 && MT <= MT_LINK0 + 16; }
// coefficients a model tag fixes at compile time (0: a run-time count)
constexpr int mt_ncoef(int MT)
{
    return mt_is_link(MT) ? MT - MT_LINK0 : ((MT >= 1 && MT <= 16) ? MT : 0);
}

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
    // tiled plan through the matrix cores (mc3hip_plan.tile_tree = 2): the operand rows with
    // the weights folded in -- [nmodel + 1][npad]: column k of the design matrix times w, then
    // the data times w (each product rounded once, on the host) -- so that a residual leaves
    // the contraction finished and the epilogue is one fused multiply-add per value
    const double *aw;
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
    // declared cadence of a run sharded over GPUs (mc3hip_problem.zlag): the other chains are
    // seen zlag refreshes later than in the strict lockstep (0)
    int32_t zlag;
    // test hook (MC3HIP_TEST_REMOTE_DELAY_NS): rows bound for another rank's replica are held
    // back this many ticks of the 100 MHz s_memrealtime clock (0 in production)
    int32_t remote_delay, _pad_delay;
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
    int32_t rs_ntp;               // ntiles rounded up to whole 128-byte lines
    int32_t rs_sched_rows;        // rs_sched: rows per column of chains
    // wavelet likelihood, abscissae on np.linspace's grid (x_i = fl(fl(i*step) + x0), last
    // point = wl_xlast; checked bit for bit at creation): they are recomputed instead of
    // stored, which frees the LDS for the data values
    int32_t wl_grid;
    int32_t wl_nreal;       // wavelet: points that are data (<= ndata; the rest is zero padding)
    double wl_x0, wl_step, wl_xlast;
    const double *data_t;         // wavelet: the data as [j][t] (thread t's j-th point)
    double *wl_scratch;           // wavelet, ndata > 2^14: [blocks][3 ndata / 4] pre-level values
    // RCCL exchange inside a hipGraph: slot `roff` of this ring receives the rows of the
    // refresh a graph node produced (fixed addresses, so the captured all-gather can be
    // replayed); nullptr = off
    double *xring;
    // several chains per workgroup (run_chain MC): chains a workgroup advances per refresh
    // (0 = one workgroup per chain) and where their waiting state starts in dynamic LDS (in
    // doubles); set per launch by the host
    int32_t mc_chains, mc_off;
    // chain-wave wavelet kernel: the hand-structured chain wave serves this problem
    // (kernels_wavelet.h lean_chain_wave, host::cw_lean_eligible)
    int32_t cw_lean;              // (bit 0: the shape; bit 1: out-of-bounds turns ride)
    // resident tiled plans: the hand-structured owner serves this problem (resident.h
    // lean_resident_owner, host::rs_lean_eligible)
    int32_t rs_lean;
    // resident_async_kernel: the blocks of sixteen chains a column's scheduler wave has made up,
    // in the order every tile wave of the column takes them -- [columns][rs_sched_rows][16]
    // words (rs_gen << 32 | chains of the block << 8 | chain of the slot); a word of another
    // launch's rs_gen = not there yet
    int32_t rs_gen;
    unsigned long long *rs_sched;
    // ... and which round of proposals of which launch a chain's proposal row belongs to:
    // [nlocal] words (rs_gen << 32 | rounds published so far in this launch), stored by the
    // owner behind the row.  The scheduler takes chain c into round R only when the word says
    // so: a row of parity R & 1 that is still there from round R - 2 (the owner resets it
    // when it collects R - 2's sums -- and nothing else orders a column's scheduler behind
    // that, since the columns' chains change from round to round) is not mistaken for R's.
    unsigned long long *rs_tag;
    // test hook (MC3HIP_TEST_SLOW_COLUMN = "column:ticks", read at creation): the tile waves
    // of that column of resident_async_kernel wait `ticks` of the 100 MHz clock before every
    // block -- the columns then drift apart as far as the hand-off allows (0 ticks: off)
    int32_t test_slow_col, test_slow_ticks;
};

constexpr int PH_FUSED = 0, PH_PROPOSE = 1, PH_FINISH = 2, PH_RESIDENT = 3, PH_CHAINWAVE = 4;
// the phases in which ONE wave runs the chain and the data term comes back through a hook
constexpr bool phase_hooked(int phase) { return phase == PH_RESIDENT || phase == PH_CHAINWAVE; }
constexpr int RS_ROW = 16;          // doubles per published proposal row: one 128-byte line
constexpr int RA_CMAX = 128;        // resident_async_kernel: chains of a column (two per lane of a
                                    // tile wave), owner waves of a workgroup at most
constexpr int RA_OWNERS = 3;        // (a twelfth wave, in the workgroups of tile 0: the scheduler)
// LDS ahead of the owners' areas, in doubles: the waves' row totals of a chain, [2][RA_CMAX + 1][8]
// (two sets of slots, by the round's parity: a column's chains change from round to round;
// row RA_CMAX: nobody's), the counters [2][RA_CMAX + 1], the blocks made up and the tile
// waves' progress (144), the give-up word (2)
constexpr int RA_PART = 2 * (RA_CMAX + 1) * 8, RA_CNT = RA_CMAX + 1;
constexpr int RA_HEAD = RA_PART + RA_CNT + 144 + 2;
constexpr int RS_WCAP = 256;        // chains whose group totals a workgroup collects in LDS
                                    // before it adds them up (resident tiled plan)
constexpr int SC_STRIDE = MC3HIP_MAX_PARAMS + 16;   // doubles per sc_nextp row (zero padded)
// in-launch hand-off (handoff.h): a double that still holds this pattern has not been
// published yet; ctl[CTL_ABORT] != 0: a wait timed out, the launch is giving up
constexpr unsigned long long ROW_PENDING = 0xffffffffffffffffull;
constexpr int CTL_ABORT = 2;                   // ctl[2] != 0: a wait timed out

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
// (threads the LDS areas were carved for: a chain-wave workgroup has one wave more)
#define MC3_LSTAMP_THREADS (blockDim.x > 512 ? 512 : blockDim.x)
// (the thread that stamps: the first of the wave that runs the pyramid's tail -- wave 1 in a
// chain-wave workgroup, wavelet_body TW)
#define MC3_LSTAMP_TID (blockDim.x > 512 ? 64u : 0u)
#define MC3_LSTAMP(slot)                                                                  \
    do {                                                                                  \
        /* (the builtin, not inline asm with an SGPR output: some stamps sit in control    \
           flow the compiler considers divergent) */                                      \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                       \
        if (threadIdx.x == MC3_LSTAMP_TID && blockIdx.x == 0)                             \
            reinterpret_cast<unsigned long long *>(mc3_smem_stamps_)[MC3_LSTAMP_THREADS + \
                MC3HIP_MAX_PARAMS + 2 + 32 + (slot)] = t_;                                \
    } while (0)
#define MC3_LSTAMP_FLUSH()                                                                \
    do {                                                                                  \
        if (threadIdx.x == MC3_LSTAMP_TID && blockIdx.x == 0)                             \
            for (int q_ = 0; q_ < 16; ++q_)                                               \
                P.stamps[q_] = reinterpret_cast<unsigned long long *>(                    \
                    mc3_smem_stamps_)[MC3_LSTAMP_THREADS + MC3HIP_MAX_PARAMS + 2 + 32 + q_]; \
    } while (0)
#endif
#ifdef MC3HIP_TICKS_ONLY
// (the draining stamps perturb what the tick sums measure: a ticks-only diagnostic build)
#undef MC3_STAMP
#define MC3_STAMP(slot) do { } while (0)
#endif
#else
#define MC3_STAMP(slot) do { } while (0)
#define MC3_TICK(var) do { } while (0)
#define MC3_LSTAMP(slot) do { } while (0)
#define MC3_LSTAMP_FLUSH() do { } while (0)
#endif

// operand / result of v_mfma_f64_16x16x4_f64
typedef double mc3_v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double make_double(int lo, int hi) { return __hiloint2double(hi, lo); }

// A launch argument (pointer to global memory) as a scalar of its own: the empty asm detaches
// it from the wide register tuple the kernel arguments arrive in (which the allocator spills
// and reloads whole), the address-space type keeps what the optimiser knew about it -- a
// laundered plain pointer would be accessed with flat instructions.
template <class T>
__device__ __forceinline__ T *own_global(T *p)
{
    typedef T __attribute__((address_space(1))) *G;
    G g = (G)p;
    asm volatile("; own" : "+s"(g));
    return (T *)g;
}


// ---- model evaluation ------------------------------------------------------
template <int MT>
struct Model {
    static constexpr int NC = mt_ncoef(MT);
    double cf[NC > 0 ? NC : 1];
    const double *coef;     // LDS copy of the proposal (generic / linear)
    const double *indep;
    int nmodel, ndata;      // ndata = column stride of A (the padded length)

    __device__ __forceinline__ void init(const DevProblem &P, const double *s_nextp)
    {
        coef = s_nextp;
        indep = P.indep;
        nmodel = P.nmodel;
        ndata = P.npad;
        if constexpr (NC > 0) {
#pragma unroll
            for (int j = 0; j < NC; ++j)
                cf[j] = s_nextp[j];
        }
    }
    // coefficients straight from wave 0's lanes (lane j holds parameter j): the
    // fast path of fully-free models, no LDS round trip
    __device__ __forceinline__ void init_from_lanes(const DevProblem &P, double npk)
    {
        coef = nullptr;
        indep = P.indep;
        nmodel = P.nmodel;
        ndata = P.npad;
        if constexpr (NC > 0) {
#pragma unroll
            for (int j = 0; j < NC; ++j)
                cf[j] = make_double(__builtin_amdgcn_readlane(__double2loint(npk), j),
                                    __builtin_amdgcn_readlane(__double2hiint(npk), j));
        }
    }
    // (lane k holds FREE parameter k; coefficient j is free parameter kf[j], or -- kf[j] < 0
    // -- the fixed value fx[j])
    __device__ __forceinline__ void init_from_lanes(const DevProblem &P, double npk,
                                                    const int (&kf)[NC > 0 ? NC : 1],
                                                    const double (&fx)[NC > 0 ? NC : 1])
    {
        coef = nullptr;
        indep = P.indep;
        nmodel = P.nmodel;
        ndata = P.npad;
        if constexpr (NC > 0) {
#pragma unroll
            for (int j = 0; j < NC; ++j) {
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
        if constexpr (NC > 0 && !mt_is_link(MT)) {
            double m = cf[NC - 1];
#pragma unroll
            for (int j = NC - 2; j >= 0; --j)
                m = mc3_fma(m, x, cf[j]);
            return m;
        } else {
            double m = coef[nmodel - 1];
            for (int j = nmodel - 2; j >= 0; --j)
                m = mc3_fma(m, x, coef[j]);
            return m;
        }
    }
    // a design-matrix model from one point's column values (MT_LINK0 + K: a register tile),
    // the terms in the order of at() below
    __device__ __forceinline__ double from_columns(const double (&a)[NC > 0 ? NC : 1]) const
    {
        double m = a[0] * cf[0];
#pragma unroll
        for (int k = 1; k < NC; ++k)
            m = mc3_fma(a[k], cf[k], m);
        return m;
    }
    __device__ __forceinline__ double at(int i) const
    {
        if constexpr (mt_is_link(MT)) {
            double m = indep[i] * cf[0];
#pragma unroll
            for (int k = 1; k < NC; ++k)
                m = mc3_fma(indep[(size_t)k * ndata + i], cf[k], m);
            return m;
        } else if constexpr (MT != MT_LINEAR) {
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
// lane l receives lane (l + 1) mod 64's value: DPP wave rotate (gfx9), VALU only
__device__ __forceinline__ double wave_rol1(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x134, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x134, 0xf, 0xf, false);
    return make_double(lo, hi);
}
// lane l receives lane src's value (any src: the LDS crossbar, no LDS memory, no barrier)
__device__ __forceinline__ double lane_gather(double v, int src)
{
    const int lo = __builtin_amdgcn_ds_bpermute(4 * src, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(4 * src, __double2hiint(v));
    return make_double(lo, hi);
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

// three such sums at once: the same additions, interleaved, so that the pipeline works on three
// independent chains instead of waiting on one (a single wave hides no latency by itself)
__device__ __forceinline__ void wave_sum3(double &a, double &b, double &c)
{
    a = add_xor32(a); b = add_xor32(b); c = add_xor32(c);
    a = add_xor16(a); b = add_xor16(b); c = add_xor16(c);
    a = add_ror<8>(a); b = add_ror<8>(b); c = add_ror<8>(c);
    a = add_ror<4>(a); b = add_ror<4>(b); c = add_ror<4>(c);
    a = add_ror<2>(a); b = add_ror<2>(b); c = add_ror<2>(c);
    a = add_ror<1>(a); b = add_ror<1>(b); c = add_ror<1>(c);
}

// The same sums for T = 2..4 chains at once, transposed: the stride-32 and stride-16 levels pair
// the lanes of TWO (FOUR) chains per instruction and leave chain q's partial sums in its own
// rows of 16 lanes, where the last four levels run for all chains together -- the pairing of
// wave_sum, level for level, so every total is bit for bit wave_sum's.  Row r of the result
// holds the total of chain r (T = 3: rows 2 and 3 both chain 2; T = 2: rows 0-1 chain 0,
// rows 2-3 chain 1).
__device__ __forceinline__ double pair_sum32(double x, double y)
{
    // lanes 0..31: x[l] + x[l + 32]; lanes 32..63: y[l - 32] + y[l]
    const auto a = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y),
                                                    false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y),
                                                    false, false);
    return make_double(a[0], b[0]) + make_double(a[1], b[1]);
}
__device__ __forceinline__ double pair_sum16(double x, double y)
{
    // rows 0, 2: x[l] + x[l + 16]; rows 1, 3: y[l - 16] + y[l]
    const auto a = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y),
                                                    false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y),
                                                    false, false);
    return make_double(a[0], b[0]) + make_double(a[1], b[1]);
}
template <int T>
__device__ __forceinline__ double rows_sum(const double (&a)[T])
{
    static_assert(T >= 2 && T <= 4, "2..4 chains");
    double t;
    if constexpr (T == 2)
        t = add_xor16(pair_sum32(a[0], a[1]));
    else
        t = pair_sum16(pair_sum32(a[0], a[2]), pair_sum32(a[1], a[T - 1]));
    t = add_ror<8>(t);
    t = add_ror<4>(t);
    t = add_ror<2>(t);
    t = add_ror<1>(t);
    return t;
}

// a double that is the same in every lane, in scalar registers from here on
__device__ __forceinline__ double uniform_double(double v)
{
    const unsigned long long b = mc3_bits(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return mc3_from_bits(((unsigned long long)hi << 32) | lo);
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
    static constexpr int NA = mt_is_link(MT) ? mt_ncoef(MT) : 1;
    double x[mt_is_link(MT) ? 1 : NV], d[NV], w[NV];
    double a[mt_is_link(MT) ? NV : 1][NA];     // MT_LINK0 + K: this thread's points of the K columns

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
            if constexpr (mt_is_link(MT)) {
#pragma unroll
                for (int k = 0; k < NA; ++k)
                    a[v][k] = P.indep[(size_t)k * P.npad + i];
            } else if constexpr (MT != MT_LINEAR) {
                x[v] = P.indep[i];
            }
        }
    }
    // a tile beyond the data (the loops with several tiles in the registers of several waves):
    // weight 0, and values the model is finite at -- every point adds fma(0, 0, acc) = acc
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            d[v] = w[v] = 0.0;
            if constexpr (mt_is_link(MT)) {
#pragma unroll
                for (int k = 0; k < NA; ++k)
                    a[v][k] = 0.0;
            } else {
                x[v] = 0.0;
            }
        }
    }
};

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

// smallest m with 2^m >= n (one instruction: the loop this was ran once per likelihood in the
// wavelet kernels, fourteen taken branches at 2^14 points)
__device__ __forceinline__ int ilog2(int n)
{
    return n <= 1 ? 0 : 32 - __builtin_clz((unsigned)(n - 1));
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

}  // namespace mc3hip
