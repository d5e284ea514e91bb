// k_lean7.hip -- the peer-to-peer flavours of the rows loops (kernels_rows.h<..., P2P>: an ensemble
// sharded over GPUs, two to four chains per wave on each); a translation unit of its own so that
// the library builds in parallel.
#include "k_rows_impl.h"

namespace mc3hip {
namespace host {

chain_kernel_fn lean_rows_instance_p2p(const mc3hip_ctx *c) { return lean_rows_pick<true>(c); }

}  // namespace host
}  // namespace mc3hip
