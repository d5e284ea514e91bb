// k_lean4.hip -- peer-to-peer instances of the hand-laid-out snooker and random-walk loops
// (kernels_lean.h lean_snooker_kernel / lean_mrw_kernel, P2P: rows into the local replica, a
// courier wave carries them to the other ranks' -- or, without it, the chain wave itself); a
// translation unit of its own so that the library builds in parallel.
#include "ctx.h"
#include "kernels_lean.h"

namespace mc3hip {
namespace host {
namespace {

template <int MT>
chain_kernel_fn lean_other_p2p_instance_t(const mc3hip_ctx *c)
{
    const bool courier = lean_courier(c, true);
    const bool fx = !lean_plain_shape(c), thin = c->P.thinning > 1;
#define MC3_LEAN_P2P(KERNEL)                                                        \
    if (thin)                                                                       \
        return courier ? KERNEL<MT, false, true, false, true, true>                 \
                       : KERNEL<MT, false, true, false, true, false>;               \
    if (fx)                                                                         \
        return courier ? KERNEL<MT, true, false, false, true, true>                 \
                       : KERNEL<MT, true, false, false, true, false>;               \
    return courier ? KERNEL<MT, false, false, false, true, true>                    \
                   : KERNEL<MT, false, false, false, true, false>;
    if (lean_mrw_eligible(c)) {
        MC3_LEAN_P2P(lean_mrw_kernel)
    }
    if (lean_snooker_eligible(c)) {
        MC3_LEAN_P2P(lean_snooker_kernel)
    }
#undef MC3_LEAN_P2P
    return nullptr;
}

}  // namespace

// the peer-to-peer kernel of a context that one of these loops serves, or nullptr
chain_kernel_fn lean_other_p2p_instance(const mc3hip_ctx *c)
{
    switch (c->mt) {
    case MT_POLY2: return lean_other_p2p_instance_t<MT_POLY2>(c);
    case MT_POLY3: return lean_other_p2p_instance_t<MT_POLY3>(c);
    default: return lean_other_p2p_instance_t<MT_POLY4>(c);
    }
}

}  // namespace host
}  // namespace mc3hip
