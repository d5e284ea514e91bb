// k_wavelet_w8cw.hip -- the nine-wave wavelet kernel (wavelet_cw_kernel, kernels_wavelet.h): a
// translation unit of its own, compiled beside k_wavelet_w8.hip (the eight-wave kernels); the
// ninth wave runs the hand-structured chain (lean_chain_wave).
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
    // (the snooker update and the random walk run the pipelined kernel, k_wavelet_w8pp.hip, also
    // with ONE chain per workgroup -- round 4; should its grid not fit, the general chain as
    // chain wave serves them)
    return c->P.sampler == MC3HIP_DEMC ? wavelet_cw_kernel<MT, MC3HIP_DEMC>
                                       : persist_instance_wavelet_w8cg(c);
}

}  // namespace

chain_kernel_fn persist_instance_wavelet_w8cw(const mc3hip_ctx *c)
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
