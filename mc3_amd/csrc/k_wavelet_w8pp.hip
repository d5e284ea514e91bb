// k_wavelet_w8pp.hip -- the wavelet kernel with several chains per workgroup PIPELINED: ten waves
// (wavelet_pipe_kernel, kernels_wavelet.h), snooker and mrw.  A translation unit of its own, as
// the other chain-wave kernels.
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
    return c->P.sampler == MC3HIP_SNOOKER ? wavelet_pipe_kernel<MT, MC3HIP_SNOOKER>
                                          : wavelet_pipe_kernel<MT, MC3HIP_MRW>;
}

}  // namespace

chain_kernel_fn persist_instance_wavelet_w8pp(const mc3hip_ctx *c)
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
