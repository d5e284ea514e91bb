// k_tiled.hip -- instances of the tiled plan's three launches per iteration (tiled.h):
// proposal, per-tile data sums of all chains (vector unit or FP64 matrix cores), finish.
#include "ctx.h"
#include "tiled.h"

namespace mc3hip {
namespace host {
namespace {

template <int MT>
void launch_propose_t(mc3hip_ctx *c, int roff, int j)
{
    MC3_LAUNCH((propose_kernel<MT>), dim3(c->P.nlocal), dim3(64), c->smem, c->stream,
                       c->P, roff, j);
}
template <int MT>
void launch_finish_t(mc3hip_ctx *c, int roff, int j)
{
    MC3_LAUNCH((finish_kernel<MT>), dim3(c->P.nlocal), dim3(64), c->smem, c->stream,
                       c->P, roff, j);
}
#define DISPATCH_MT1(FN, ...)                                                       \
    switch (c->mt) {                                                                \
    case MT_POLY2: FN<MT_POLY2>(__VA_ARGS__); break;                                \
    case MT_POLY3: FN<MT_POLY3>(__VA_ARGS__); break;                                \
    case MT_POLY4: FN<MT_POLY4>(__VA_ARGS__); break;                                \
    case MT_LINEAR: FN<MT_LINEAR>(__VA_ARGS__); break;                              \
    default: FN<MT_POLY_GENERIC>(__VA_ARGS__); break;                               \
    }

// LDS of the tile kernel: 16 chains x 1024 partial sums + two buffers of 16 coefficient rows
constexpr size_t TILE_LDS_BYTES = (16 * 1024 + 2 * 16 * (16 + 2)) * sizeof(double);

template <int KMAX, bool LINEAR, int KX = 0>
void launch_tile_t(mc3hip_ctx *c)
{
    const int ncol = (c->P.nlocal + c->tile_chain_block - 1) / c->tile_chain_block;
    constexpr int TP = KMAX <= 4 ? 4 : 2;
    MC3_LAUNCH_CHAIN(c, (tile_chisq_kernel<KMAX, LINEAR, TP, KX>), dim3(c->P.ntiles, ncol),
                       dim3(1024), TILE_LDS_BYTES, c->stream, c->P,
                       c->tile_chain_block, (const double *)c->P.sc_nextp,
                       (const int32_t *)c->P.sc_flag);
}

template <int KB>
void launch_tile_mfma_t(mc3hip_ctx *c)
{
    const int ncol = (c->P.nlocal + c->tile_chain_block - 1) / c->tile_chain_block;
    MC3_LAUNCH_CHAIN(c, (tile_chisq_mfma_kernel<KB>), dim3(c->P.ntiles, ncol), dim3(512),
                       0, c->stream, c->P, c->tile_chain_block,
                       (const double *)c->P.sc_nextp, (const int32_t *)c->P.sc_flag);
}

}  // namespace

void launch_propose(mc3hip_ctx *c, int roff, int j) { DISPATCH_MT1(launch_propose_t, c, roff, j) }
void launch_finish(mc3hip_ctx *c, int roff, int j) { DISPATCH_MT1(launch_finish_t, c, roff, j) }

hipError_t tile_set_attr(mc3hip_ctx *c)
{
    const int bytes = (int)TILE_LDS_BYTES;
    hipError_t e = hipSuccess;
#define TILE_ATTR(KMAX, LIN)                                                               \
    e = hipFuncSetAttribute(                                                               \
        reinterpret_cast<const void *>(&tile_chisq_kernel<KMAX, LIN, (KMAX <= 4 ? 4 : 2)>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    const bool lin = c->mt == MT_LINEAR;
#define TILE_ATTR_X(KX)                                                                    \
    e = hipFuncSetAttribute(                                                               \
        reinterpret_cast<const void *>(&tile_chisq_kernel<4, false, 4, KX>),               \
        hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    switch (tile_exact_degree(c)) {
    case 2: TILE_ATTR_X(2) return e;
    case 3: TILE_ATTR_X(3) return e;
    case 4: TILE_ATTR_X(4) return e;
    default: break;
    }
#undef TILE_ATTR_X
    switch (c->tile_kmax) {
    case 4: if (lin) { TILE_ATTR(4, true) } else { TILE_ATTR(4, false) } break;
    // (design matrices of 2..15 columns go through the matrix cores: tile_uses_mfma; the
    // vector-unit tile kernel keeps the one- and the sixteen-column instances)
    // (should one of those ever come this way: the sixteen-column instance holds them too)
    case 8: if (lin) { TILE_ATTR(16, true) } else { TILE_ATTR(8, false) } break;
    case 12: if (lin) { TILE_ATTR(16, true) } else { TILE_ATTR(12, false) } break;
    default: if (lin) { TILE_ATTR(16, true) } else { TILE_ATTR(16, false) } break;
    }
#undef TILE_ATTR
    return e;
}

void launch_tile(mc3hip_ctx *c)
{
    if (tile_uses_mfma(c)) {
        switch ((c->P.nmodel + 4) / 4) {             // operand rows: the coefficients + the data
        case 1: launch_tile_mfma_t<1>(c); break;
        case 2: launch_tile_mfma_t<2>(c); break;
        case 3: launch_tile_mfma_t<3>(c); break;
        default: launch_tile_mfma_t<4>(c); break;
        }
        return;
    }
    switch (tile_exact_degree(c)) {
    case 2: launch_tile_t<4, false, 2>(c); return;
    case 3: launch_tile_t<4, false, 3>(c); return;
    case 4: launch_tile_t<4, false, 4>(c); return;
    default: break;
    }
    const bool lin = c->mt == MT_LINEAR;
    switch (c->tile_kmax) {
    case 4: lin ? launch_tile_t<4, true>(c) : launch_tile_t<4, false>(c); break;
    case 8: lin ? launch_tile_t<16, true>(c) : launch_tile_t<8, false>(c); break;
    case 12: lin ? launch_tile_t<16, true>(c) : launch_tile_t<12, false>(c); break;
    default: lin ? launch_tile_t<16, true>(c) : launch_tile_t<16, false>(c); break;
    }
}

}  // namespace host
}  // namespace mc3hip
