// k_lean5.hip -- instances of the hand-laid-out DEMC loop with two or three chains per wave and two tiles
// (kernels_lean.h lean_demc_mc_kernel) and with helper waves (lean_demc_wide_kernel, an
// experiment behind MC3HIP_LEAN_WAVES); a translation unit of its own so that the library
// builds in parallel.
#include "ctx.h"
#include "kernels_lean.h"

namespace mc3hip {
namespace host {
namespace {

template <int MT>
chain_kernel_fn lean_demc_mc_instance_t(const mc3hip_ctx *c, bool p2p)
{
    // (2 049 ... 16 384 points: the tiles in the registers of four / eight waves per chain)
    if (!p2p && lean_demc_eligible(c) && c->P.fepsilon == 0.0 && lean_plain_shape(c)) {
        switch (wide_tiles(c)) {
        case 2: return lean_demc_wide_kernel<MT, 4, 2>;
        case 4: return lean_demc_wide_kernel<MT, 4, 4>;
        case 8: return lean_demc_wide_kernel<MT, 4, 8>;
        case 16: return lean_demc_wide_kernel<MT, 8, 16>;
        default: break;
        }
    }
    // (two tiles, two or three chains per wave; the one-tile shapes: kernels_rows.h, k_lean6.hip)
    if (lean_demc_mc_eligible(c) && !p2p) {
        if (c->mc_chains == 2)
            return lean_demc_mc_kernel<MT, 2, false, true>;
        return lean_demc_mc_kernel<MT, 3, false, true>;
    }
    return nullptr;
}

}  // namespace

chain_kernel_fn lean_demc_mc_instance(const mc3hip_ctx *c, bool p2p)
{
    switch (c->mt) {
    case MT_POLY2: return lean_demc_mc_instance_t<MT_POLY2>(c, p2p);
    case MT_POLY3: return lean_demc_mc_instance_t<MT_POLY3>(c, p2p);
    default: return lean_demc_mc_instance_t<MT_POLY4>(c, p2p);
    }
}

}  // namespace host
}  // namespace mc3hip
