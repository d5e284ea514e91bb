// k_wavelet_w8mc.hip -- the nine-wave wavelet kernel with several chains per workgroup
// (wavelet_cw_kernel<.., MCK = true>, kernels_wavelet.h): a translation unit of its own, so that
// it compiles beside k_wavelet_w8.hip and the single-chain kernel there keeps its layout.
#include "ctx.h"
#include "kernels_fused.h"
#include "wavelet.h"
#include "kernels_wavelet.h"

namespace mc3hip {
namespace host {
namespace {

template <int MT>
chain_kernel_fn instance_t(const mc3hip_ctx *c)
{
    // (several chains of the snooker update or the random walk per workgroup: the pipelined
    // kernel, ctx.h chainwave_pipe)
    return c->P.sampler == MC3HIP_DEMC ? wavelet_cw_kernel<MT, MC3HIP_DEMC, true> : nullptr;
}

}  // namespace

chain_kernel_fn persist_instance_wavelet_w8mc(const mc3hip_ctx *c)
{
    switch (c->mt) {
    case MT_POLY2: return instance_t<MT_POLY2>(c);
    case MT_POLY3: return instance_t<MT_POLY3>(c);
    case MT_POLY4: return instance_t<MT_POLY4>(c);
    case MT_LINEAR: return instance_t<MT_LINEAR>(c);
    default: return instance_t<MT_POLY_GENERIC>(c);
    }
}

}  // namespace host
}  // namespace mc3hip
