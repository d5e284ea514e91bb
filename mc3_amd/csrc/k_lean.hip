// k_lean.hip -- instances of the lean one-wave persistent kernels (kernels_lean.h): the
// hand-laid-out DEMC loop and the lean instances of run_chain, per polynomial degree, sampler,
// peer-to-peer / fixed-coefficient / streamed-tile flavour.
#include "ctx.h"
#include "kernels_lean.h"

namespace mc3hip {
namespace host {
namespace {

// Several chains per wave taking TURNS (run_chain MC; one GPU): what is left to the lean
// instances of run_chain since the hand-laid-out loops took over the one-chain-per-wave shapes
// (round 3) and two to four chains side by side (kernels_rows.h, round 4) -- ensembles of more
// than four chains per wave, thinned or streamed-tile shapes with more chains than waves.
template <int MT, int SMP>
chain_kernel_fn fast_instance_mc(const mc3hip_ctx *c)
{
    if (c->P.npad > VLANES)
        return persist_fast_kernel<MT, SMP, false, 1, false, true, true>;
    if (c->P.nfree != c->P.nparams || c->P.ngprior > 0 || c->P.nshare > 0)
        return persist_fast_kernel<MT, SMP, false, 1, true, false, true>;
    return c->P.thinning == 1 ? persist_fast_kernel<MT, SMP, false, 1, false, false, true>
                              : persist_fast_kernel<MT, SMP, false, 0, false, false, true>;
}

template <int MT, int SMP>
chain_kernel_fn fast_instance_tt(const mc3hip_ctx *c, bool p2p)
{
    // (one chain per wave, or peer to peer: a hand-laid-out loop serves every shape
    // fast_eligible lets through -- nothing here)
    return (c->mc_chains > 0 && !p2p) ? fast_instance_mc<MT, SMP>(c) : nullptr;
}

template <int MT>
chain_kernel_fn fast_instance_t(const mc3hip_ctx *c, bool p2p)
{
    // (the loops of the declared multi-GPU cadence: k_lean3.hip)
    if (chain_kernel_fn f = lean_lag_instance(c, p2p))
        return f;
    // (2..4 chains per wave side by side, plain shapes: k_lean6.hip)
    if (chain_kernel_fn f = lean_rows_instance(c, p2p))
        return f;
    // (the hand-laid-out snooker, random-walk and thinned DEMC loops: k_lean2.hip; their
    // peer-to-peer flavours with a courier wave: k_lean4.hip)
    if (chain_kernel_fn f = p2p ? lean_other_p2p_instance(c) : lean_other_instance(c))
        return f;
    if (c->P.sampler == MC3HIP_MRW)
        return fast_instance_tt<MT, MC3HIP_MRW>(c, p2p);
    // (2..4 chains per wave, helper waves: k_lean5.hip)
    if (chain_kernel_fn f = lean_demc_mc_instance(c, p2p))
        return f;
    if (lean_demc_eligible(c)) {
        if (c->P.fepsilon != 0.0)                        // (one GPU, one tile: fast_eligible)
            return lean_plain_shape(c) ? lean_demc_kernel<MT, false, false, false, false, false, true>
                                       : lean_demc_kernel<MT, false, true, false, false, false, true>;
        if (p2p)
            return lean_courier(c, p2p) ? lean_demc_kernel<MT, true, false, false, false, true>
                                        : lean_demc_kernel<MT, true>;
        if (c->P.npad > VLANES)
            return lean_demc_kernel<MT, false, false, true>;
        if (c->P.nfree != c->P.nparams || c->P.ngprior > 0 || c->P.nshare > 0)
            return lean_demc_kernel<MT, false, true>;
        // the plain shape: the draws of 64 refreshes at once (BATCH).  Written for ensembles of
        // more than one wave per SIMD; since the row is published where it is decided (round 6)
        // it is the faster flavour at every size -- 64 / 256 / 512 / 1024 chains: 57.9 / 227.1 /
        // 438.8 / 814.7 M chain-steps/s against 56.4 / 222.5 / 431.7 / 773.8 with the draws made
        // refresh by refresh (profiles/r06_handoff_variants.txt) -- and the only one built
        // (more than 1 024 chains -- two and more waves per SIMD --: the wait laid out by hand,
        // kernels_lean.h BIG)
        return c->P.nlocal > 1024 ? lean_demc_kernel<MT, false, false, false, true, false, false, true>
                                  : lean_demc_kernel<MT, false, false, false, true>;
    }
    if (c->P.sampler == MC3HIP_DEMC)
        return fast_instance_tt<MT, MC3HIP_DEMC>(c, p2p);
    return fast_instance_tt<MT, MC3HIP_SNOOKER>(c, p2p);
}

}  // namespace

chain_kernel_fn fast_instance(const mc3hip_ctx *c, bool p2p)
{
    switch (c->mt) {
    case MT_POLY2: return fast_instance_t<MT_POLY2>(c, p2p);
    case MT_POLY3: return fast_instance_t<MT_POLY3>(c, p2p);
    default: return fast_instance_t<MT_POLY4>(c, p2p);
    }
}

}  // namespace host
}  // namespace mc3hip
