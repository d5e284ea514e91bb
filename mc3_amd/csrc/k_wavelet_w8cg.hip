// k_wavelet_w8cg.hip -- the nine-wave wavelet kernel whose ninth wave runs the GENERAL chain
// (wavelet_cw_kernel<.., false, true>: trace, priors, shared parameters, thinning).
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
    return c->P.sampler == MC3HIP_SNOOKER ? wavelet_cw_kernel<MT, MC3HIP_SNOOKER, false, true>
           : c->P.sampler == MC3HIP_DEMC  ? wavelet_cw_kernel<MT, MC3HIP_DEMC, false, true>
                                          : wavelet_cw_kernel<MT, MC3HIP_MRW, false, true>;
}

}  // namespace

chain_kernel_fn persist_instance_wavelet_w8cg(const mc3hip_ctx *c)
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
