// k_general_impl.h -- instances of the general chain kernels with the Gaussian likelihood for
// ONE workgroup width: one workgroup of MC3_WPC (1, 4 or 16) waves per chain, one launch per
// refresh (step_kernel), many refreshes per launch (persist_kernel, also with peer-to-peer
// stores), chisq of arbitrary parameter vectors (eval_chisq_kernel).  Included by
// k_general_w{1,4,16}.hip, which are compiled in parallel.
#include "ctx.h"
#include "kernels_fused.h"

#define MC3_CAT_(a, b) a##b
#define MC3_CAT(a, b) MC3_CAT_(a, b)
#define MC3_FN(name) MC3_CAT(name, MC3_WPC)

namespace mc3hip {
namespace host {
namespace {

constexpr int WPC = MC3_WPC;

// (four waves per chain are chosen for more than 4 096 points only -- mc3hip.hip choose_wpc --
// so their instances are the several-tile ones)
constexpr bool ONE_TILE = WPC == 1;

template <int MT>
void launch_step_t(mc3hip_ctx *c, int roff)
{
    if (!ONE_TILE || c->P.npad > VLANES) {
        MC3_LAUNCH_CHAIN(c, (step_kernel<WPC, MT, false, true>), dim3(c->P.nlocal), dim3(64 * WPC),
                           c->smem, c->stream, c->P, roff);
    } else if constexpr (ONE_TILE) {
        MC3_LAUNCH_CHAIN(c, (step_kernel<WPC, MT, false, false>), dim3(c->P.nlocal),
                           dim3(64 * WPC), c->smem, c->stream, c->P, roff);
    }
}

// the exact instance (tile count, peer-to-peer stores) that will be launched
template <int MT>
void persist_instance_t(const mc3hip_ctx *c, bool p2p, chain_kernel_fn *out)
{
    const bool mc = c->mc_chains > 0 && !p2p;          // several chains per workgroup
    if (!ONE_TILE || c->P.npad > VLANES) {
        *out = p2p ? persist_kernel<WPC, MT, false, true, true>
                   : mc ? persist_kernel<WPC, MT, false, true, false, true>
                        : persist_kernel<WPC, MT, false, true>;
    } else if constexpr (ONE_TILE) {
        *out = p2p ? persist_kernel<WPC, MT, false, false, true>
                   : mc ? persist_kernel<WPC, MT, false, false, false, true>
                        : persist_kernel<WPC, MT, false, false>;
    }
}

template <int MT>
void launch_eval_t(mc3hip_ctx *c, int n)
{
    MC3_LAUNCH((eval_chisq_kernel<WPC, MT, false>), dim3(n), dim3(64 * WPC), c->smem,
                       c->stream, c->P, (const double *)c->eval_params, c->eval_out);
}

#define DISPATCH_MT(FN, ...)                                                        \
    switch (c->mt) {                                                                \
    case MT_POLY2: FN<MT_POLY2>(__VA_ARGS__); break;                                \
    case MT_POLY3: FN<MT_POLY3>(__VA_ARGS__); break;                                \
    case MT_POLY4: FN<MT_POLY4>(__VA_ARGS__); break;                                \
    case MT_LINEAR: FN<MT_LINEAR>(__VA_ARGS__); break;                              \
    default: FN<MT_POLY_GENERIC>(__VA_ARGS__); break;                               \
    }

}  // namespace

void MC3_FN(launch_step_gauss_w)(mc3hip_ctx *c, int roff) { DISPATCH_MT(launch_step_t, c, roff) }

chain_kernel_fn MC3_FN(persist_instance_gauss_w)(const mc3hip_ctx *c, bool p2p)
{
    chain_kernel_fn f = nullptr;
    DISPATCH_MT(persist_instance_t, c, p2p, &f)
    return f;
}

void MC3_FN(launch_eval_gauss_w)(mc3hip_ctx *c, int n) { DISPATCH_MT(launch_eval_t, c, n) }

}  // namespace host
}  // namespace mc3hip
