// k_lean6.hip -- instances of the hand-laid-out loops with the two to four chains of a wave
// side by side in its rows of 16 lanes (kernels_rows.h): one-tile shapes of ensembles with more chains than waves can be co-resident, one
// GPU or sharded over several (P2P); a translation unit of its own so that the library builds in parallel.
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
        return fx ? lean_demc_rows_kernel<MT, T, true, P2P>
                  : lean_demc_rows_kernel<MT, T, false, P2P>;
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

}  // namespace

// the kernel of a context that one of these loops serves, or nullptr
chain_kernel_fn lean_rows_instance(const mc3hip_ctx *c, bool p2p)
{
    if (!lean_rows_eligible(c))
        return nullptr;
    switch (c->mt) {
    case MT_POLY2: return p2p ? lean_rows_instance_t<MT_POLY2, true>(c)
                              : lean_rows_instance_t<MT_POLY2, false>(c);
    case MT_POLY3: return p2p ? lean_rows_instance_t<MT_POLY3, true>(c)
                              : lean_rows_instance_t<MT_POLY3, false>(c);
    default: return p2p ? lean_rows_instance_t<MT_POLY4, true>(c)
                        : lean_rows_instance_t<MT_POLY4, false>(c);
    }
}

}  // namespace host
}  // namespace mc3hip
