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
    // ONE chain per workgroup, snooker without the support distribution (C4): the chain wave
    // with its state in registers and its draws worked out a batch ahead (pipe_chain_wave_one)
    // (not for the generic polynomial, whose compute waves leave the chain wave no registers)
    if constexpr (MT != MT_POLY_GENERIC) {
        if (c->P.sampler == MC3HIP_SNOOKER && c->mc_chains == 1 && c->P.fepsilon == 0.0 &&
            !env_text("MC3HIP_NO_CW_ONE"))
            return wavelet_pipe_kernel<MT, MC3HIP_SNOOKER, true>;
    }
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
