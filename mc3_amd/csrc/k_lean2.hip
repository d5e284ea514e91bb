// k_lean2.hip -- instances of the hand-laid-out snooker and random-walk loops (one chain per
// wave, thinning, streamed tiles; kernels_lean.h); a translation
// unit of its own so that the library builds in parallel.
#include "ctx.h"
#include "kernels_lean.h"

namespace mc3hip {
namespace host {
namespace {

template <int MT>
chain_kernel_fn lean_other_instance_t(const mc3hip_ctx *c)
{
    if (lean_mrw_eligible(c)) {
        if (c->P.thinning > 1)
            return lean_mrw_kernel<MT, false, true>;
        if (c->P.npad > VLANES)
            return lean_mrw_kernel<MT, false, false, true>;
        return lean_plain_shape(c) ? lean_mrw_kernel<MT> : lean_mrw_kernel<MT, true>;
    }
    if (lean_snooker_eligible(c)) {
        // (2 049 ... 16 384 points: the tiles in the registers of four / eight waves per chain)
        if (c->P.fepsilon == 0.0 && c->P.thinning == 1 && lean_plain_shape(c)) {
            switch (wide_tiles(c)) {
            case 2: return lean_snooker_wide_kernel<MT, 4, 2>;
            case 4: return lean_snooker_wide_kernel<MT, 4, 4>;
            case 8: return lean_snooker_wide_kernel<MT, 4, 8>;
            case 16: return lean_snooker_wide_kernel<MT, 8, 16>;
            default: break;
            }
        }
        if (c->P.fepsilon != 0.0)                   // (thinning 1, one tile: fast_eligible)
            return lean_plain_shape(c)
                       ? lean_snooker_kernel<MT, false, false, false, false, false, true>
                       : lean_snooker_kernel<MT, true, false, false, false, false, true>;
        if (c->P.thinning > 1)
            return lean_snooker_kernel<MT, false, true>;
        if (c->P.npad > VLANES)
            return lean_snooker_kernel<MT, false, false, true>;
        return lean_plain_shape(c) ? lean_snooker_kernel<MT> : lean_snooker_kernel<MT, true>;
    }
    return nullptr;
}

}  // namespace

// the kernel of a context that one of these loops serves (one GPU), or nullptr
chain_kernel_fn lean_other_instance(const mc3hip_ctx *c)
{
    switch (c->mt) {
    case MT_POLY2: return lean_other_instance_t<MT_POLY2>(c);
    case MT_POLY3: return lean_other_instance_t<MT_POLY3>(c);
    default: return lean_other_instance_t<MT_POLY4>(c);
    }
}

}  // namespace host
}  // namespace mc3hip
