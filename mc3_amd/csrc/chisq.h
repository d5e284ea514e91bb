// chisq.h -- chisq of a parameter vector: data term (Gaussian or wavelet) + priors.
#pragma once
#include "core.h"
#include "gauss.h"

namespace mc3hip {

// (defined in wavelet.h, which only the wavelet kernels' translation unit includes)
template <int WPC, int MT>
__device__ __forceinline__ void wavelet_stage(const DevProblem &P, double *xs);
// (NX: one wave, the noise terms left in s_tab by a helper wave of the workgroup -- a
// workgroup barrier where they are first used: kernels_wavelet.h wavelet_one_kernel)
template <int WPC, int MT, bool LARGE = false, bool NX = false>
__device__ double block_wavelet(const DevProblem &P, const double *s_nextp, double *s_part,
                                int *s_flag, double *s_tab, double *s_halo, const double *xs);

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

// chisq of a chain from the per-tile sums plus the prior term.  The tiles' sums are added as
// the wave's halving tree, sixty-four at a time (a lane per tile, lanes beyond the tiles 0.0;
// groups of sixty-four in order) -- the order of ResidentOwner::collect and of the checker
// (oracle_lockstep.c tiles_sum): six dependent additions for C5's 49 tiles where adding
// them one after the other was 49, each behind whatever the FP64 pipe is busy with.  The +inf-model trap of chain.py:322-323 is resolved lazily: a +inf model
// value always makes the sum non-finite (w > 0), so only a non-finite sum needs
// the scan that tells "+inf model" (chisq = inf) from any other overflow.
template <int MT>
__device__ __forceinline__ double finish_chisq(const DevProblem &P, int lc)
{
    const double *np = P.sc_nextp + (size_t)lc * SC_STRIDE;
    const int lane = threadIdx.x & 63;
    double total = 0.0;
    for (int t0 = 0; t0 < P.ntiles; t0 += 64) {
        const double v = (t0 + lane < P.ntiles) ? P.sc_part[(size_t)lc * P.ntiles + t0 + lane]
                                                : 0.0;
        total = total + wave_sum(v);
    }
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

}  // namespace mc3hip
