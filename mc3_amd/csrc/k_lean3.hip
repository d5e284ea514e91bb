// k_lean3.hip -- instances of the hand-laid-out loops of the declared multi-GPU cadence
// (kernels_lean.h lean_demc_lag_kernel: zlag >= 1) and of the peer-to-peer flavours with a
// courier wave; a translation unit of its own so that the library builds in parallel.
#include "ctx.h"
#include "kernels_lean.h"

namespace mc3hip {
namespace host {
namespace {

template <int MT>
chain_kernel_fn lean_lag_instance_t(const mc3hip_ctx *c, bool p2p)
{
    if (!lean_demc_lag_eligible(c))
        return nullptr;
    const bool fx = !lean_plain_shape(c);
#define MC3_LAG(P2P_, COUR_)                                                        \
    return fx ? lean_demc_lag_kernel<MT, P2P_, true, COUR_>                         \
              : lean_demc_lag_kernel<MT, P2P_, false, COUR_>;
    if (p2p) {
        if (lean_courier(c, true)) {
            MC3_LAG(true, true)
        }
        MC3_LAG(true, false)
    }
    MC3_LAG(false, false)
#undef MC3_LAG
}

}  // namespace

// the kernel of a context in the declared multi-GPU cadence that one of these loops serves
// (one GPU or peer to peer), or nullptr
chain_kernel_fn lean_lag_instance(const mc3hip_ctx *c, bool p2p)
{
    switch (c->mt) {
    case MT_POLY2: return lean_lag_instance_t<MT_POLY2>(c, p2p);
    case MT_POLY3: return lean_lag_instance_t<MT_POLY3>(c, p2p);
    default: return lean_lag_instance_t<MT_POLY4>(c, p2p);
    }
}

}  // namespace host
}  // namespace mc3hip
