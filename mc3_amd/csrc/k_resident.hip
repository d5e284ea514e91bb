// k_resident.hip -- instances of the tiled plan as ONE persistent launch (resident.h): design
// matrices through the FP64 matrix cores, polynomials of 2..4 coefficients on the vector unit.
#include "ctx.h"
#include "resident.h"

namespace mc3hip {
namespace host {
namespace {

// LDS: row totals [RS_WCAP][16], the give-up word, one run_chain area per wave, the last block
// of operand rows [16][512]
constexpr size_t RESIDENT_OWN = 64 + MC3HIP_MAX_PARAMS + 2 + 48 + 4;
constexpr size_t RESIDENT_LDS_BYTES =
    (RS_WCAP * 16 + 2 + 8 * RESIDENT_OWN + 16 * 512) * sizeof(double);
// (only four operand blocks need the last term)
constexpr size_t RESIDENT_LDS_BYTES_REGS = (RS_WCAP * 16 + 2 + 8 * RESIDENT_OWN) * sizeof(double);
// polynomials (ResidentPoly): two exchange buffers [16][512], the give-up word, the areas
constexpr size_t RESIDENT_POLY_LDS_BYTES = (2 * 16 * 512 + 2 + 8 * RESIDENT_OWN) * sizeof(double);

// resident_async_kernel: the waves' row totals, counters, blocks made up, the give-up word
// (core.h RA_HEAD), the owners' areas, the operand blocks beyond two
constexpr size_t RESIDENT_ASYNC_LDS_DOUBLES = RA_HEAD + RA_OWNERS * RESIDENT_OWN;

}  // namespace

size_t resident_async_lds_bytes(const mc3hip_ctx *c)
{
    // (operand blocks beyond two -- beside the general chain as owner: beyond one, up to three
    // blocks in all -- live in LDS, [16][512] each)
    const int kb = (c->P.nmodel + 4) / 4;
    const int kr = (resident_owner_lean(c) || kb == 4) ? 2 : 1;
    return (RESIDENT_ASYNC_LDS_DOUBLES + (kb > kr ? (kb - kr) * 16 * 512 : 0)) * sizeof(double);
}

size_t resident_lds_bytes(const mc3hip_ctx *c)
{
    if (resident_poly(c))
        return RESIDENT_POLY_LDS_BYTES;
    return (c->P.nmodel + 4) / 4 == 4 ? RESIDENT_LDS_BYTES : RESIDENT_LDS_BYTES_REGS;
}

namespace {
template <bool P2P, bool LEAN>
resident_kernel_fn pick(const mc3hip_ctx *c)
{
    if (resident_poly(c)) {
        switch (c->P.nmodel) {
        case 2: return resident_poly_kernel<2, P2P, LEAN>;
        case 3: return resident_poly_kernel<3, P2P, LEAN>;
        default: return resident_poly_kernel<4, P2P, LEAN>;
        }
    }
    switch ((c->P.nmodel + 4) / 4) {          // operand rows: coefficients + the data column
    case 1: return resident_tile_kernel<1, P2P, LEAN>;
    case 2: return resident_tile_kernel<2, P2P, LEAN>;
    case 3: return resident_tile_kernel<3, P2P, LEAN>;
    default: return resident_tile_kernel<4, P2P, LEAN>;
    }
}
}  // namespace

namespace {
template <bool P2P, bool LEAN>
resident_kernel_fn pick_async(const mc3hip_ctx *c)
{
    switch ((c->P.nmodel + 4) / 4) {
    case 1: return resident_async_kernel<1, P2P, LEAN>;
    case 2: return resident_async_kernel<2, P2P, LEAN>;
    case 3: return resident_async_kernel<3, P2P, LEAN>;
    default:
        // (four blocks of operand rows beside the general chain as owner: the kernel in step
        // serves those -- ctx.h resident_async_shape)
        if constexpr (LEAN)
            return resident_async_kernel<4, P2P, true>;
        else
            return nullptr;
    }
}
}  // namespace

resident_kernel_fn resident_async_kernel_of(const mc3hip_ctx *c, bool p2p)
{
    const bool lean = resident_owner_lean(c);
    return p2p ? (lean ? pick_async<true, true>(c) : pick_async<true, false>(c))
               : (lean ? pick_async<false, true>(c) : pick_async<false, false>(c));
}

// (the owner -- hand-structured or general -- is a function of the problem alone, so the
// occupancy query that sizes the grid and the launch agree: ctx.h resident_owner_lean)
resident_kernel_fn resident_kernel(const mc3hip_ctx *c, bool p2p)
{
    const bool lean = resident_owner_lean(c);
    return p2p ? (lean ? pick<true, true>(c) : pick<true, false>(c))
               : (lean ? pick<false, true>(c) : pick<false, false>(c));
}

}  // namespace host
}  // namespace mc3hip
