// k_rows_impl.h -- choosing the instance of the rows loops (kernels_rows.h) for a context; included
// by k_lean6.hip (one GPU) and k_lean7.hip (sharded over GPUs, peer to peer): two translation
// units so that the library builds in parallel.
#pragma once
#include "ctx.h"
#include "kernels_rows.h"

namespace mc3hip {
namespace host {
namespace {

template <int MT, int T, bool P2P>
chain_kernel_fn lean_rows_instance_tt(const mc3hip_ctx *c)
{
    // (FX: fixed or shared coefficients, Gaussian priors)
    const bool fx = !lean_plain_shape(c);
    switch (c->P.sampler) {
    case MC3HIP_DEMC:
        if (c->P.zlag > 0)                   // (the declared cadence: rows asked for a refresh ahead)
            return fx ? lean_demc_rows_lag_kernel<MT, T, true, P2P>
                      : lean_demc_rows_lag_kernel<MT, T, false, P2P>;
        // (strict DEMC with fixed coefficients or priors has no peer-to-peer loop at all --
        // ctx.h fast_eligible -- so it has no rows loop either)
        if constexpr (P2P) {
            return fx ? nullptr : lean_demc_rows_kernel<MT, T, false, true>;
        } else {
            return fx ? lean_demc_rows_kernel<MT, T, true, false>
                      : lean_demc_rows_kernel<MT, T, false, false>;
        }
    case MC3HIP_SNOOKER:
        return fx ? lean_snooker_rows_kernel<MT, T, true, P2P>
                  : lean_snooker_rows_kernel<MT, T, false, P2P>;
    default:
        return fx ? lean_mrw_rows_kernel<MT, T, true, P2P>
                  : lean_mrw_rows_kernel<MT, T, false, P2P>;
    }
}

template <int MT, bool P2P>
chain_kernel_fn lean_rows_instance_t(const mc3hip_ctx *c)
{
    switch (c->mc_chains) {
    case 2: return lean_rows_instance_tt<MT, 2, P2P>(c);
    case 3: return lean_rows_instance_tt<MT, 3, P2P>(c);
    default: return lean_rows_instance_tt<MT, 4, P2P>(c);
    }
}

template <bool P2P>
chain_kernel_fn lean_rows_pick(const mc3hip_ctx *c)
{
    switch (c->mt) {
    case MT_POLY2: return lean_rows_instance_t<MT_POLY2, P2P>(c);
    case MT_POLY3: return lean_rows_instance_t<MT_POLY3, P2P>(c);
    default: return lean_rows_instance_t<MT_POLY4, P2P>(c);
    }
}

}  // namespace
}  // namespace host
}  // namespace mc3hip
