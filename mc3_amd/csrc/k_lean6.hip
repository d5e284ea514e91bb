// k_lean6.hip -- instances of the hand-laid-out loops with the two to four chains of a wave
// side by side in its rows of 16 lanes (kernels_rows.h): one-tile shapes of ensembles with more
// chains than waves can be co-resident, on one GPU (the peer-to-peer flavours: k_lean7.hip).
#include "k_rows_impl.h"

namespace mc3hip {
namespace host {

// the kernel of a context that one of these loops serves, or nullptr
chain_kernel_fn lean_rows_instance(const mc3hip_ctx *c, bool p2p)
{
    if (!lean_rows_eligible(c))
        return nullptr;
    return p2p ? lean_rows_instance_p2p(c) : lean_rows_pick<false>(c);
}

}  // namespace host
}  // namespace mc3hip
