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

// (SMP: the sampler, known at compile time -- the chain wave's path between two likelihoods is
// what an iteration waits for, so the other samplers' branches are compiled out of it)
template <int MT, int SMP>
__global__ __launch_bounds__(576) void wavelet_cw_kernel(const DevProblem P, const int nref)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SmemLayout L = carve(smem, P, 512);           // the compute waves' areas
    double *own = L.xs + (P.ndata > 128 ? P.ndata : 128);
    int *s_cmd = reinterpret_cast<int *>(own + CW_OWN);
    double *s_result = own + CW_OWN + 2;
    const int tid = threadIdx.x;
    // a launch queued behind one that gave up must not run: its inputs are missing
    if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(
            &P.ctl[CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
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
        run_chain<1, MT, false, false, true, SMP, false, PH_CHAINWAVE, false, 0, false,
                  WaveletHook>(P, P.ctl[0], nref, 0, -1, &H, (int)blockIdx.x, own);
        if (!H.stopped) {                               // the run is over: the others leave too
            if ((tid & 63) == 0)
                s_cmd[0] = CW_STOP;
            block_sync<9>();
        }
        return;
    }
    // ---- the compute waves -------------------------------------------------------------------
    for (;;) {
        block_sync<9>();                                // (A)
        MC3_LSTAMP(0);
        // (the order and the coefficients in one trip to LDS, not one after the other)
        const int cmd = s_cmd[0];
        Model<MT> model;
        model.init(P, L.nextp);
        if (cmd == CW_STOP)
            return;
        if (cmd == CW_SKIP)
            continue;
        double c;
        if (B == 32)
            c = wavelet_body<8, MT, 32, false, true>(P, L.nextp, L.part, L.flag, L.tab, L.halo,
                                                     L.xs, nullptr, 0, nullptr, 0, &model);
        else
            c = wavelet_body<8, MT, 16, false, true>(P, L.nextp, L.part, L.flag, L.tab, L.halo,
                                                     L.xs, nullptr, 0, nullptr, 0, &model);
        if (tid == 0)
            s_result[0] = c;
        MC3_LSTAMP(1);
        block_sync<9>();                                // (3)
        MC3_LSTAMP(2);
        MC3_LSTAMP_FLUSH();
    }
}

}  // namespace mc3hip
