// kernels_fused.h -- one workgroup per chain: per-refresh and persistent kernels, chisq evaluation.
#pragma once
#include "run_chain.h"

namespace mc3hip {

template <int WPC, int MT, bool WL, bool MULTI>
__global__ __launch_bounds__(64 * WPC) void step_kernel(const DevProblem P, const int roff)
{
    run_chain<WPC, MT, WL, MULTI, false>(P, P.ctl[0] + roff, 1);
    ring_out(P, roff);
}

// many Z refreshes per launch with in-launch hand-off of the new rows; every
// workgroup of the grid must be resident (the host checks before choosing this path)
// (MC: more chains than co-resident workgroups -- a workgroup advances several, run_chain MC)
template <int WPC, int MT, bool WL, bool MULTI, bool P2P = false, bool MC = false>
__global__ __launch_bounds__(64 * WPC) void persist_kernel(const DevProblem P, const int nref)
{
    run_chain<WPC, MT, WL, MULTI, true, -1, false, PH_FUSED, P2P, 0, false, NoHook, MC>(
        P, P.ctl[0], nref);
}

// chisq for arbitrary parameter vectors (Chain.eval_model(ret='chisq'))
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

}  // namespace mc3hip
