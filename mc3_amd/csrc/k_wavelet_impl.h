// k_wavelet_impl.h -- instances of the general chain kernels with the wavelet likelihood (the
// blocked D4 pyramid, wavelet.h) for ONE geometry: MC3_WW = 1, 4 or 8 waves per chain (a
// function of ndata), or -- MC3_WL_LARGE -- the large-data instances (more than 2^14 points:
// eight waves, pre-levels through global memory).  Included by k_wavelet_{w1,w4,w8,large}.hip,
// which are compiled in parallel.
#include "ctx.h"
#include "kernels_fused.h"
#include "wavelet.h"
#include "kernels_wavelet.h"   // (CW_EXTRA)

#define MC3_CAT_(a, b) a##b
#define MC3_CAT(a, b) MC3_CAT_(a, b)
#define MC3_FN(name) MC3_CAT(name, MC3_WL_TAG)

namespace mc3hip {
namespace host {
namespace {

#ifdef MC3_WL_LARGE
constexpr int WW = 8;
constexpr bool LARGE = true;
#else
constexpr int WW = MC3_WW;
constexpr bool LARGE = false;
#endif

template <int MT>
void launch_step_t(mc3hip_ctx *c, int roff)
{
    MC3_LAUNCH_CHAIN(c, (step_kernel<WW, MT, true, LARGE>), dim3(c->P.nlocal), dim3(64 * WW),
                       c->smem, c->stream, c->P, roff);
}

template <int MT>
void persist_instance_t(const mc3hip_ctx *c, bool p2p, chain_kernel_fn *out)
{
    const bool mc = c->mc_chains > 0 && !p2p;          // several chains per workgroup
    if constexpr (WW == 8 && !LARGE) {
        static_assert(CW_EXTRA == CW_EXTRA_DOUBLES, "LDS of the chain wave");
        if (chainwave_runs(c, p2p)) {                   // a wave of its own for the chain
            // (the nine-wave kernels are compiled in units of their own: k_wavelet_w8cw.hip, the
            // hand-structured chain wave; _w8cg, the general chain; _w8mc, several chains per
            // workgroup)
            // (... and _w8pp, the same pipelined over the chains of a workgroup: ten waves)
            *out = chainwave_pipe(c, p2p) ? persist_instance_wavelet_w8pp(c)
                   : c->mc_chains > 0 ? persist_instance_wavelet_w8mc(c)
                   : c->P.cw_lean   ? persist_instance_wavelet_w8cw(c)
                                    : persist_instance_wavelet_w8cg(c);
            return;
        }
    }
    if constexpr (WW == 8 && !LARGE) {
        // (the fused eight-wave kernel in which wave 0 was also the chain went with round 4:
        // every instance of it spilled.  Where the nine-wave kernel does not run -- the
        // peer-to-peer mode -- there is no persistent launch: one launch per refresh, and the
        // multi-GPU exchange through RCCL or the host)
        (void)mc;
        *out = nullptr;
    } else {
        *out = p2p ? persist_kernel<WW, MT, true, LARGE, true>
                   : mc ? persist_kernel<WW, MT, true, LARGE, false, true>
                        : persist_kernel<WW, MT, true, LARGE>;
        // (one wave per chain, snooker, the common shape: the chain written out by hand)
        // (... and four: 4 096 points)
        if constexpr ((WW == 1 || WW == 4) && !LARGE) {
            if (!p2p && !mc && c->wl_one)
                *out = c->P.sampler == MC3HIP_SNOOKER ? wavelet_one_kernel<MT, WW>
                       : c->P.sampler == MC3HIP_MRW
                           ? wavelet_one_other_kernel<MT, MC3HIP_MRW, WW>
                           : wavelet_one_other_kernel<MT, MC3HIP_DEMC, WW>;
        }
    }
}

template <int MT>
void launch_eval_t(mc3hip_ctx *c, int n)
{
    MC3_LAUNCH((eval_chisq_kernel<WW, MT, true, LARGE>), dim3(n), dim3(64 * WW), c->smem,
                       c->stream, c->P, (const double *)c->eval_params, c->eval_out);
}

// kernels that need more than the default 64 KiB of dynamic LDS must opt in
template <int MT>
void set_attr_t(mc3hip_ctx *c, hipError_t *e)
{
    *e = hipFuncSetAttribute(reinterpret_cast<const void *>(&step_kernel<WW, MT, true, LARGE>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->smem);
    if (*e == hipSuccess)
        *e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&eval_chisq_kernel<WW, MT, true, LARGE>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->smem);
}

// MC3_MT_SPLIT (the one-wave geometry: five block sizes x five model kinds made this the
// longest unit of the build): this unit holds the polynomials of 2..4 coefficients, a sister
// unit -- the same file with MC3_MT_SECOND, tag MC3_WL_TAG2 -- design matrices and the general
// degree; the public functions of this unit forward what is not theirs.
#if defined(MC3_MT_SECOND)
#define DISPATCH_MT(FN, ...)                                                        \
    switch (c->mt) {                                                                \
    case MT_LINEAR: FN<MT_LINEAR>(__VA_ARGS__); break;                              \
    default: FN<MT_POLY_GENERIC>(__VA_ARGS__); break;                               \
    }
#elif defined(MC3_MT_SPLIT)
#define DISPATCH_MT(FN, ...)                                                        \
    switch (c->mt) {                                                                \
    case MT_POLY2: FN<MT_POLY2>(__VA_ARGS__); break;                                \
    case MT_POLY3: FN<MT_POLY3>(__VA_ARGS__); break;                                \
    default: FN<MT_POLY4>(__VA_ARGS__); break;                                      \
    }
#else
#define DISPATCH_MT(FN, ...)                                                        \
    switch (c->mt) {                                                                \
    case MT_POLY2: FN<MT_POLY2>(__VA_ARGS__); break;                                \
    case MT_POLY3: FN<MT_POLY3>(__VA_ARGS__); break;                                \
    case MT_POLY4: FN<MT_POLY4>(__VA_ARGS__); break;                                \
    case MT_LINEAR: FN<MT_LINEAR>(__VA_ARGS__); break;                              \
    default: FN<MT_POLY_GENERIC>(__VA_ARGS__); break;                               \
    }
#endif
#if defined(MC3_MT_SPLIT) && !defined(MC3_MT_SECOND)
#define MC3_FN2(name) MC3_CAT(name, MC3_WL_TAG2)
#define MC3_NOT_MINE(c) ((c)->mt != MT_POLY2 && (c)->mt != MT_POLY3 && (c)->mt != MT_POLY4)
#define MC3_FORWARD(call) if (MC3_NOT_MINE(c)) return call;
#else
#define MC3_FORWARD(call)
#endif

}  // namespace

void MC3_FN(launch_step_wavelet_)(mc3hip_ctx *c, int roff)
{
    MC3_FORWARD(MC3_FN2(launch_step_wavelet_)(c, roff))
    DISPATCH_MT(launch_step_t, c, roff)
}

chain_kernel_fn MC3_FN(persist_instance_wavelet_)(const mc3hip_ctx *c, bool p2p)
{
    MC3_FORWARD(MC3_FN2(persist_instance_wavelet_)(c, p2p))
    chain_kernel_fn f = nullptr;
    DISPATCH_MT(persist_instance_t, c, p2p, &f)
    return f;
}

void MC3_FN(launch_eval_wavelet_)(mc3hip_ctx *c, int n)
{
    MC3_FORWARD(MC3_FN2(launch_eval_wavelet_)(c, n))
    DISPATCH_MT(launch_eval_t, c, n)
}

hipError_t MC3_FN(set_smem_attr_wavelet_)(mc3hip_ctx *c)
{
    MC3_FORWARD(MC3_FN2(set_smem_attr_wavelet_)(c))
    hipError_t e = hipSuccess;
    DISPATCH_MT(set_attr_t, c, &e)
    return e;
}

}  // namespace host
}  // namespace mc3hip
