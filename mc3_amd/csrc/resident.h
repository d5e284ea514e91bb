// resident.h -- the tiled plan as ONE persistent launch.
#pragma once
#include "core.h"
#include "handoff.h"
#include "run_chain.h"

namespace mc3hip {

// -DMC3HIP_TIMELINE=R (with MC3HIP_STAMPS): who did what when in iteration R of a launch, on the
// 100 MHz clock all CUs share -- P.stamps[1024 + 4 c + k]: chain c's owner (partner rows there,
// proposal published, sums collected, row published); [4096 + 16 col + b]: the scheduler
// published block b; [8192 + 32 wg + 4 b + k]: tile wave 0 of a workgroup (block b taken,
// rows there, products begun, sums left); [16384 + c]: the block chain c was in
#ifdef MC3HIP_TIMELINE
#define MC3_TL(cond, slot)                                                                \
    do {                                                                                  \
        if (cond)                                                                         \
            P.stamps[slot] = __builtin_amdgcn_s_memrealtime();                            \
    } while (0)
#else
#define MC3_TL(cond, slot) do { } while (0)
#endif

// ---- resident tiled plan --------------------------------------------------------------
// The tiled plan as ONE launch (design-matrix models through the matrix cores; one GPU, or
// one launch per GPU with the Z rows published to every rank).  The grid is the tile kernel's
// -- tiles x columns of chains, one workgroup per CU -- but every workgroup keeps its tile
// (operand rows in registers and LDS, weights in LDS) for the whole launch instead of
// reloading 200 KB of it per iteration, and the three phases of an iteration hand over
// inside the launch instead of at kernel boundaries:
//   * a wave that OWNS a chain (wave v of workgroup w owns chain w + v * workgroups) runs the
//     ordinary run_chain for it -- proposal, Metropolis, Z append -- and publishes the proposed
//     coefficient row (rs_prop, one 128-byte line, by iteration parity);
//   * all eight waves of every workgroup then compute the data sums of their tile for the
//     chains of their column, 16 chains at a time, as soon as those rows are there, each wave
//     at its own pace; after the last block one barrier, the sums of the waves are added and
//     one sum per (chain, tile) is published (rs_part);
//   * the owner adds the sums of its chain in tile order, hands the slots and the consumed row
//     back (all-ones), and goes on to the next iteration.
// Every handed-over double is its own flag (all-ones = not there yet), exactly as for Z rows;
// no grid barrier, no fence.  Why a row may be reused: a slot is handed back only after all
// the sums that were computed from it have arrived, and a workgroup asks for the rows of
// iteration i + 2 only after it has used those of i + 1, which the owner published after
// handing back those of i.  The data column rides in the contraction as operand row K with
// coefficient -1 (D = model - data, the same single rounding as the subtraction it replaces).
// Same arithmetic and the same summation order as tile_chisq_mfma_kernel: the oracle does not
// know which of the two ran.
// What the owner of a chain does around the tile phase, whatever computes the tile sums
// (ResidentTile: design matrices through the matrix cores; ResidentPoly: polynomials on the
// vector unit).
struct ResidentOwner {
    int lane;
    int par;                // parity of the iteration whose proposals are used next
    bool failed;            // a wait of this wave timed out: said at the next agreement point
    // owner waves: their LDS area ([0..63]: the Gaussian priors as index, centre, lower and
    // upper width, 16 each)
    double *s_own;

    // this wave owns chain lc; `area` is its run_chain LDS area (whose first 64 doubles
    // run_chain does not use here)
    __device__ __forceinline__ void set_owner(const DevProblem &P, int lc, double *area)
    {
        s_own = area;
        if (P.ngprior <= 16 && lane < P.ngprior) {
            const int jp = P.igprior[lane];
            area[lane] = (double)jp;
            area[16 + lane] = P.prior[jp];
            area[32 + lane] = P.priorlow[jp];
            area[48 + lane] = P.priorup[jp];
        }
    }

    // owner, before the tile phase: the proposed coefficients of chain lc for everybody
    __device__ __forceinline__ void publish(const DevProblem &P, int lc, const double *s_nextp)
    {
        // (the slots handed back during the previous iteration have landed before the row
        // that lets anybody write them again leaves)
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
        if (lane < RS_ROW) {
            const int K = P.nmodel;
            double v = (lane < K) ? s_nextp[lane] : ((lane == K) ? -1.0 : 0.0);
            if (mc3_bits(v) == ROW_PENDING)
                v = mc3_from_bits(0x7ff8000000000000ull);
            __hip_atomic_store(&P.rs_prop[((size_t)par * P.nlocal + lc) * RS_ROW + lane], v,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // owner, after the tile phase: the data sum of chain lc, the tiles' sums added as the wave's
    // halving tree; hands the slots and the row this iteration used back
    __device__ __forceinline__ double collect(const DevProblem &P, int lc)
    {
        long long *ctl = const_cast<long long *>(P.ctl);
        double *row = P.rs_part + (size_t)lc * P.rs_ntp;
        double total = 0.0;
        for (int t0 = 0; t0 < P.ntiles; t0 += 64) {
            const int n = (P.ntiles - t0 < 64) ? P.ntiles - t0 : 64;
            double *mine = row + t0 + (lane < n ? lane : 0);
            const double *const addr[1] = {mine};
            double val[1] = {load_published(mine)};
            if (!wait_rows<1>(addr, val, ctl))
                failed = true;
            // (the wave's halving tree over the tiles, lanes beyond them 0.0: chisq.h
            // finish_chisq's order)
            total = total + wave_sum(lane < n ? val[0] : 0.0);
            if (lane < n)
                __hip_atomic_store(mine, mc3_from_bits(ROW_PENDING), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane < RS_ROW)
            __hip_atomic_store(&P.rs_prop[((size_t)(par ^ 1) * P.nlocal + lc) * RS_ROW + lane],
                               mc3_from_bits(ROW_PENDING), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        return total;
    }

    // the chisq of the proposal from that sum (finish_chisq of the three-launch plan)
    template <int MT>
    __device__ __forceinline__ double finish(const DevProblem &P, double total,
                                             const double *s_nextp) const
    {
        const bool finite = (mc3_bits(total) & 0x7ff0000000000000ull) != 0x7ff0000000000000ull;
        if (!finite) {
            Model<MT> model;
            model.init(P, s_nextp);
            int saw = 0;
            for (int i = lane; i < P.ndata; i += 64)
                saw |= is_pos_inf(model.at(i));
            if (__any(saw))
                return mc3_from_bits(0x7ff0000000000000ull);
        }
        if (is_pos_inf(total))
            return total;
        if (P.ngprior > 16)
            return total + prior_term(P, s_nextp);
        // prior_term from the LDS copy of its constants: lane g works out term g (the
        // divisions side by side instead of one after the other), then the terms are added
        // in order
        double term = 0.0;
        if (lane < P.ngprior) {
            const int jp = (int)s_own[lane];
            const double d = s_nextp[jp] - s_own[16 + lane];
            const double sd = (d > 0.0) ? d / s_own[48 + lane] : d / s_own[32 + lane];
            term = sd * sd;
        }
        double pr = 0.0;
        for (int gp = 0; gp < P.ngprior; ++gp)
            pr = pr + read_lane(term, gp);
        return total + pr;
    }
};

// resident_async_kernel: the owner is a wave of its own -- nothing of the tile phase is its
// business (what run_chain calls through its hook at that point only turns the parity over)
struct ResidentOwnerAsync : ResidentOwner {
    // (rs_gen << 32 | rounds of proposals published so far in this launch: DevProblem::rs_tag)
    unsigned long long tag;
    // the proposal row, and behind it the word that says which round's it is.  The word may
    // overtake the row on its way: whoever reads it knows that the row's elements were reset
    // before either left (publish() waits for the resets of collect() to land first), and an
    // element that is not there yet says so itself.
    __device__ __forceinline__ void publish(const DevProblem &P, int lc, const double *s_nextp)
    {
        ResidentOwner::publish(P, lc, s_nextp);
        tag += 1;
        if (lane == 0)
            __hip_atomic_store(&P.rs_tag[lc], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ bool tile_phase(const DevProblem &, bool alive_in)
    {
        // (a wait that timed out, or the test hook, has told everybody: ctl[CTL_ABORT])
        if (!alive_in || failed)
            return false;
        par ^= 1;
        return true;
    }
};

template <int KB, int KRMAX = 3>
struct ResidentTile : ResidentOwner {
    // operand rows of the tile: blocks of four rows in registers (up to three: 96 registers);
    // a fourth block (12..15 coefficients + the data) lives in LDS, 64 KB, one read per product
    // (KRMAX = 2, resident_async_kernel: two in registers, the others in LDS)
    static constexpr int KR = (KB > KRMAX) ? KRMAX : KB;
    static constexpr int KL = KB - KR;              // blocks in LDS
    double bop[8][2][KR];
    const double *s_b;      // LDS [KL][16][512]: this thread's operands of row blocks KR..
    double *s_w;            // LDS [RS_WCAP][16]: row totals of the chains done so far ([8..15] = 0)
    int *s_stop;            // LDS: a wave of this workgroup has given up
    int t, c0, c1, wv;

    __device__ __forceinline__ void load(const DevProblem &P, int tile, int col, int chain_block,
                                         double *lds_w, int *lds_stop, double *lds_b)
    {
        s_b = lds_b + threadIdx.x;
        s_w = lds_w;
        s_stop = lds_stop;
        t = tile;
        c0 = col * chain_block;
        c1 = (c0 + chain_block < P.nlocal) ? c0 + chain_block : P.nlocal;
        lane = threadIdx.x & 63;
        wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        par = 0;
        failed = false;
        s_own = nullptr;
        const int j = lane & 15, g = lane >> 4, K = P.nmodel;
#pragma unroll
        for (int vb = 0; vb < 8; ++vb) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const size_t i = (size_t)t * 2048 + 1024 * h + 128 * wv + 16 * vb + j;
                // (operand rows with the weights folded in, P.aw: design columns times w, then
                // the data times w as row K)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const int k = 4 * kb + g;
                    const double v = (k <= K) ? P.aw[(size_t)k * P.npad + i] : 0.0;
                    if (kb < KR)
                        bop[vb][h][kb < KR ? kb : 0] = v;
                    else
                        lds_b[((kb - KR) * 16 + 2 * vb + h) * 512 + threadIdx.x] = v;
                }
            }
        }
    }

    // every wave of the workgroup, once per iteration; false (for all of them) = give up
    __device__ __forceinline__ bool tile_phase(const DevProblem &P, bool alive_in)
    {
        if (!alive_in || failed)
            *s_stop = 1;
        block_sync<8>();
        if (*s_stop != 0)
            return false;
        long long *ctl = const_cast<long long *>(P.ctl);
        const double *prop = P.rs_prop + (size_t)par * P.nlocal * RS_ROW;
        const int j = lane & 15, g = lane >> 4;
        const double *addr[KB];
        double aop[KB];
        auto request = [&](int cb) {
            const int lc = (cb + j < c1) ? cb + j : c1 - 1;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                addr[kb] = prop + (size_t)lc * RS_ROW + 4 * kb + g;
                aop[kb] = load_published(addr[kb]);
            }
        };
        // (the weights and the LDS part of the tile are read where they are used: an offset
        // the compiler cannot see through keeps it from hoisting the reads out of the loops
        // and parking the values in registers again)
        int wo = 0;
        // Step s = 0..15 of a block of chains: half h = s & 1 of block vb = s >> 1 of 16 lanes.
        // The product of a step, with the coefficients in a[]; bl = its operand from LDS
        auto product = [&](int st, const double(&a)[KB], double bl) {
            const int vb = st >> 1, h = st & 1;
            mc3_v4d m = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                m = __builtin_amdgcn_mfma_f64_16x16x4f64(
                    a[kb], (kb < KR) ? bop[vb][h][kb < KR ? kb : 0] : bl, m, 0, 0, 0);
            return m;
        };
        auto lds_operand = [&](int st) { return (KR < KB) ? s_b[wo + st * 512] : 0.0; };
        // the 16 lanes of a row (strides 8..1; lane j = 0 has the defined order), once per block
        // of chains: a lane has added up its sixteen points of the tile by then
        auto row_total = [&](const double(&acc)[4], int off) {
            double sr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double sg = add_ror<8>(acc[r]);
                sg = add_ror<4>(sg);
                sg = add_ror<2>(sg);
                sr[r] = add_ror<1>(sg);
            }
            if (j == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s_w[(off + g + 4 * r) * 16 + wv] = sr[r];
                    s_w[(off + g + 4 * r) * 16 + 8 + wv] = 0.0;    // (the tree below is 16 wide)
                }
            }
        };
        // FP64 matrix and vector instructions share one pipe on this chip
        // (tools/probes/mfma_f64_issue_probe.hip: their times add up), so what can be hidden is
        // latency, not work.  With the weights folded into the operand rows
        // (mc3hip_plan.tile_tree = 2) the product IS the residual and the vector work of a step
        // is four fused multiply-adds: 255 registers hold the whole tile, and the compiler's own
        // schedule of the plain loop beats the hand-written software pipeline the heavier
        // epilogue needed (C5: 33.0 -> 30.4 us per iteration; profiles/r03_c5_schedules.jsonl).
        request(c0);
        int base = c0;                  // first chain of the window collected in s_w
        for (int cb = c0; cb < c1; cb += 16) {
            if (!wait_rows<KB>(addr, aop, ctl))
                failed = true;
            double a[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                a[kb] = aop[kb];
            if (cb + 16 < c1)
                request(cb + 16);              // the next block's rows travel behind this one
            asm volatile("; operands of this block" : "+v"(wo));
            double acc[4] = {0.0, 0.0, 0.0, 0.0};      // chain g + 4 r: this lane's points so far
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const mc3_v4d m = product(st, a, lds_operand(st));
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[r] = mc3_fma(m[r], m[r], acc[r]);
            }
            row_total(acc, cb - base);
            // The waves of the workgroup are NOT synchronised per block of chains: every wave
            // leaves its row totals in LDS and goes on (a barrier, an exchange and a
            // sixteen-lane sum per block cost ~10 % of the phase, and two waves that drift
            // apart share their SIMD's FP64 pipe better than two in step).  The totals are
            // added up -- eight per chain, strides 4..1 -- once per window of RS_WCAP chains,
            // normally once per iteration.
            const int end = (cb + 16 < c1) ? cb + 16 : c1;
            if (end == c1 || end - base == RS_WCAP) {
                block_sync<8>();
                for (int ch0 = base + 4 * wv; ch0 < end; ch0 += 32) {
                    const int ch = ch0 + (lane >> 4);
                    double sg = s_w[((ch < end ? ch : end - 1) - base) * 16 + (lane & 15)];
                    sg = add_ror<8>(sg);
                    sg = add_ror<4>(sg);
                    sg = add_ror<2>(sg);
                    sg = add_ror<1>(sg);
                    if (mc3_bits(sg) == ROW_PENDING)   // (a NaN that looks like "not there yet")
                        sg = mc3_from_bits(0x7ff8000000000000ull);
                    if ((lane & 15) == 0 && ch < end)
                        __hip_atomic_store(&P.rs_part[(size_t)ch * P.rs_ntp + t], sg,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (end < c1)
                    block_sync<8>();            // (the next window overwrites these totals)
                base = end;
            }
        }
        par ^= 1;
        return true;
    }

};

// The same one-launch plan for polynomials of KX = 2..4 coefficients (tile_chisq_kernel's
// arithmetic: Horner on the vector unit, 4096 points per tile, lanes first -- one halving tree
// over the tile's 1024 strided partial sums).  Eight waves: thread l owns virtual lanes l and
// l + 512 with four points each (x, data, weight: 48 registers), so the first level of the
// tree is an addition inside the thread; 16 chains at a time, their 64 coefficients one per
// lane (v_readlane hands them to the scalar unit); the 512 partial sums of a chain cross the
// workgroup through LDS (two buffers of 16 x 512: one barrier per block of chains), wave v
// finishes chains 2 v and 2 v + 1.
template <int KX>
struct ResidentPoly : ResidentOwner {
    double x[2][4], d[2][4], w[2][4];
    double *s_x;            // LDS [2][16][512]
    int *s_stop;
    int t, c0, c1, wv;

    __device__ __forceinline__ void load(const DevProblem &P, int tile, int col, int chain_block,
                                         double *lds_x, int *lds_stop)
    {
        s_x = lds_x;
        s_stop = lds_stop;
        t = tile;
        c0 = col * chain_block;
        c1 = (c0 + chain_block < P.nlocal) ? c0 + chain_block : P.nlocal;
        lane = threadIdx.x & 63;
        wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        par = 0;
        failed = false;
        s_own = nullptr;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t i = (size_t)t * 4096 + threadIdx.x + 512 * v + 1024 * q;
                x[v][q] = P.indep[i];
                d[v][q] = P.data[i];
                w[v][q] = P.w[i];
            }
        }
    }

    __device__ __forceinline__ bool tile_phase(const DevProblem &P, bool alive_in)
    {
        if (!alive_in || failed)
            *s_stop = 1;
        block_sync<8>();
        if (*s_stop != 0)
            return false;
        long long *ctl = const_cast<long long *>(P.ctl);
        const double *prop = P.rs_prop + (size_t)par * P.nlocal * RS_ROW;
        const int l = threadIdx.x;
        const double *addr[1];
        double val[1];
        auto request = [&](int cb) {
            const int lc = (cb + (lane >> 2) < c1) ? cb + (lane >> 2) : c1 - 1;
            addr[0] = prop + (size_t)lc * RS_ROW + (lane & 3);
            val[0] = load_published(addr[0]);
        };
        request(c0);
        int buf = 0;
        for (int cb = c0; cb < c1; cb += 16, buf ^= 1) {
            if (!wait_rows<1>(addr, val, ctl))
                failed = true;
            const double cfv = val[0];
            if (cb + 16 < c1)
                request(cb + 16);              // the next block's rows travel behind this one
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                double cf[KX];
#pragma unroll
                for (int k = 0; k < KX; ++k)
                    cf[k] = read_lane(cfv, 4 * g + k);
                double acc[2];
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    acc[v] = 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        double m = cf[KX - 1];            // Horner from coefficient K-1 down
#pragma unroll
                        for (int k = KX - 2; k >= 0; --k)
                            m = mc3_fma(m, x[v][q], cf[k]);
                        const double r = (m - d[v][q]) * w[v][q];
                        acc[v] = mc3_fma(r, r, acc[v]);
                    }
                }
                s_x[(buf * 16 + g) * 512 + l] = acc[0] + acc[1];       // stride 512
            }
            block_sync<8>();
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int ch = 2 * wv + cc;
                double p[8];
#pragma unroll
                for (int w2 = 0; w2 < 8; ++w2)
                    p[w2] = s_x[(buf * 16 + ch) * 512 + lane + 64 * w2];
#pragma unroll
                for (int s2 = 4; s2 >= 1; s2 >>= 1) {                 // strides 256, 128, 64
#pragma unroll
                    for (int w2 = 0; w2 < s2; ++w2)
                        p[w2] = p[w2] + p[w2 + s2];
                }
                double tot = wave_sum(p[0]);                          // strides 32..1
                if (mc3_bits(tot) == ROW_PENDING)      // (a NaN that looks like "not there yet")
                    tot = mc3_from_bits(0x7ff8000000000000ull);
                if (lane == 0 && cb + ch < c1)
                    __hip_atomic_store(&P.rs_part[(size_t)(cb + ch) * P.rs_ntp + t], tot,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // (no second barrier: the next block writes the other buffer, and this one is not
            // written again before everybody has passed the next barrier)
        }
        par ^= 1;
        return true;
    }
};

// ---- the owner of a chain, hand-structured ------------------------------------------------
// run_chain behind the hooks above carries every feature of the general chain, and with them
// a few hundred scalar registers spilled to vector lanes: of the ~11 000 cycles an owner spends
// outside the tile phase (tools/ticks.py c5) some 7 000 are its own instructions -- proposal
// 900, chisq from the sum 1 400, Metropolis 900, Z row 1 200, loop 900 -- while the other
// seven waves of its workgroup, and every workgroup of its column, wait for it.  This is the
// same chain for the commonest shape (host::rs_lean_eligible: DEMC, thinning 1, no trace, no
// shared parameters), as kernels_wavelet.h's lean_chain_wave is for the wavelet kernel: the
// next draws and log u are computed while the partner rows travel, the partners of the next
// iteration are asked for right behind the own row, and the launch arguments are scalars of
// their own.  Same arithmetic in the same order: bit-identical to run_chain's.
// ASYNC: the owner is a wave of its own (resident_async_kernel below): nothing of the tile phase
// is its business, it publishes, waits for the sums and goes on.
template <int MT, bool P2P, class H, bool ASYNC = false>
__device__ __forceinline__ void lean_resident_owner(const DevProblem &P, const int nref, H &T,
                                                    const int lc, double *area)
{
    const int lane = threadIdx.x & 63;
    const int c = P.chain0 + lc;
    double *Zr = own_global(P.Z), *log_post = own_global(P.log_post);
    double *bestx = own_global(P.bestx);
    const double *snap0 = own_global(P.snap0);
    int32_t *zchain = own_global(P.zchain);
    long long *best_iter = own_global(P.best_iter);
    long long M0 = P.M0;
    uint64_t seed = P.seed;
    double gamma = P.gamma, fepsilon = P.fepsilon;
    int nchains = P.nchains, nparams = P.nparams, nf = P.nfree, zs = P.zs;
    asm volatile("; own" : "+s"(M0), "+s"(seed));
    asm volatile("; own" : "+s"(gamma), "+s"(fepsilon), "+s"(nchains), "+s"(nparams), "+s"(nf),
                           "+s"(zs));
    double *s_nextp = area + 64;                       // (carve(): behind the one-wave `part`)
    // P2P: the ranks' replicas, their pointers read once (handoff.h PeerTable)
    PeerTable<double> peers_z, peers_lp;
    if constexpr (P2P) {
        peers_z.load(P.peerZ, P.npeers);
        peers_lp.load(P.peerLP, P.npeers);
    }
    (void)peers_z; (void)peers_lp;
    const bool mine = lane < nf;
    const int kk = mine ? lane : 0;
    const long long R0 = P.ctl[0];
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    const double *snap_first = (R0 == 0) ? snap0 : Zr + (size_t)(M0 + (R0 - 1) * nchains) * zs;
    double xk = mine ? snap_first[(size_t)c * zs + lane] : 0.0;
    double cur = P.chisq[lc];
    double best = P.best_chisq[lc];
    const long long nacc0 = P.numaccept[lc];
    long long nacc = 0;
    unsigned long long oob = 0;
    const int jfree = mine ? P.ifree[lane] : 0;
    const double pmin_k = mine ? P.fmin[lane] : 0.0;
    const double pmax_k = mine ? P.fmax[lane] : 0.0;
    const bool need_normal = fepsilon != 0.0;
    const double pstep_k = (mine && need_normal) ? P.fstep[lane] : 0.0;
    const long long abort_at = (P.test_abort >= 0 && lc == 0) ? (long long)P.test_abort : -1;
    if (lane < nparams)
        s_nextp[lane] = P.params[lane];
    if (lane + 64 < nparams)
        s_nextp[lane + 64] = P.params[lane + 64];
    auto normal_draw = [&](uint64_t it) {              // chain.py:185, as run_chain's
        double nk = 0.0;
        if (mine) {
            const Draw2 d = draw2(seed, it, (uint32_t)c, (uint32_t)(3 + (lane >> 1)));
            const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(d.q0)));
            double sn, cs;
            mc3_det_sincos2pi(u01(d.q1), &sn, &cs);
            nk = (rad * ((lane & 1) ? sn : cs)) * pstep_k;
        }
        return nk;
    };
    // DEMC's two partner rows (chain.py:221-232), of refresh R - 1: asked for while their
    // chains may still be computing them -- what is not there yet comes back as the sentinel
    // and is asked for again
    const double *ra[2] = {nullptr, nullptr};
    double rv[2] = {0.0, 0.0};
    const long long zlag = P.zlag;
    auto request_rows = [&](uint64_t q0, long long R) {
        // (zlag > 0, the declared multi-GPU cadence: the rows of refresh R - 1 - zlag)
        const long long Rs = R - 1 - zlag;
        const double *snap = (Rs < 0) ? snap0 : Zr + (size_t)(M0 + Rs * nchains) * zs;
        int r1, r2;
        demc_partners(q0, nchains, c, r1, r2);
        ra[0] = snap + (size_t)r1 * zs + kk;
        ra[1] = snap + (size_t)r2 * zs + kk;
        rv[0] = load_published<P2P>(ra[0]);
        rv[1] = load_published<P2P>(ra[1]);
    };
    Draw2 nxt = draw2(seed, (uint64_t)R0, (uint32_t)c, 0u);
    request_rows(nxt.q0, R0);
    double normal_ahead = need_normal ? normal_draw((uint64_t)R0) : 0.0;
    bool alive = true, give_up = false;
#ifdef MC3HIP_STAMPS
    unsigned long long ta8 = 0, ta10 = 0, ta11 = 0, ta12 = 0, ta13 = 0;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0, tk_end = 0;
#endif

    // (everything loaded so far retired here: see lean_chain_wave)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (long long R = R0; R < R0 + nref; ++R) {
        unsigned long long tk0 = 0, tk1 = 0, tkP = 0, tkT = 0, tkC = 0, tk2 = 0, tk3 = 0;
        (void)tk0; (void)tk1; (void)tkP; (void)tkT; (void)tkC; (void)tk2; (void)tk3;
        MC3_TICK(tk0);
#ifdef MC3HIP_STAMPS
        if (tk_end != 0)
            ta13 += tk0 - tk_end;
#endif
        if (R == abort_at) {                           // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            give_up = true;
        }
        const double u = u01(nxt.q1);                  // Metropolis draw, chain.py:259
        const double normal_k = normal_ahead;
        // in the shadow of the rows' round trip: the next draws, log u
        nxt = draw2(seed, (uint64_t)(R + 1), (uint32_t)c, 0u);
        pin_scalar(nxt.q0);
        pin_scalar(nxt.q1);
        double acc_lo, acc_hi;
        accept_window(u, acc_lo, acc_hi);
        if (need_normal)
            normal_ahead = normal_draw((uint64_t)(R + 1));
        alive = wait_rows<2, P2P>(ra, rv, ctl_rw) && alive;
        MC3_TICK(tk1);
#ifdef MC3HIP_TIMELINE
        const bool tl_on = ASYNC && R - R0 == MC3HIP_TIMELINE && lane == 0;
        MC3_TL(tl_on, 1024 + 4 * lc);
#endif
        const double jump_k = gamma * (rv[0] - rv[1]) + fepsilon * normal_k;
        // proposal and boundaries, chain.py:235-243
        const double npk = xk + jump_k;
        const bool outk = mine && (npk < pmin_k || npk > pmax_k);
        const bool anyout = __any(outk);
        if (mine)
            s_nextp[jfree] = npk;
        T.publish(P, lc, s_nextp);
        MC3_TICK(tkP);
#ifdef MC3HIP_TIMELINE
        MC3_TL(tl_on, 1024 + 4 * lc + 1);
#endif
        if constexpr (ASYNC) {
            // (a wait that timed out has told everybody: ctl[CTL_ABORT])
            if (!alive || give_up)
                break;
            T.par ^= 1;
        } else {
            alive = T.tile_phase(P, alive && !give_up);
        }
        MC3_TICK(tkT);
        if (!alive)
            break;
        // (also of a proposal that is out of bounds: the sums must be seen before the row is
        // reused)
        const double total = T.collect(P, lc);
        MC3_TICK(tkC);
#ifdef MC3HIP_TIMELINE
        MC3_TL(tl_on, 1024 + 4 * lc + 2);
#endif
        if constexpr (ASYNC) {
            if (T.failed)
                break;
        }
        bool accepted = false;
        if (!anyout) {
            const double nextchisq = T.template finish<MT>(P, total, s_nextp);
            MC3_TICK(tk2);
            accepted = metropolis_accept(0.5 * (cur - nextchisq), u, acc_lo, acc_hi);
            if (accepted) {                            // chain.py:257-274
                xk = npk;
                cur = nextchisq;
                nacc += 1;
            }
        } else {
            oob += outk ? 1ull : 0ull;                 // chain.py:242-243
        }
        MC3_TICK(tk3);
        // Z append, chain.py:276-289
        const long long row = M0 + R * nchains + c;
        if constexpr (P2P) {
            store_row_everywhere(P, peers_z, (size_t)row * zs + lane, xk, mine);   // publish
        } else if (mine) {
            __hip_atomic_store(&Zr[(size_t)row * zs + lane], xk, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);           // publish (sc1)
        }
#ifdef MC3HIP_TIMELINE
        MC3_TL(tl_on, 1024 + 4 * lc + 3);
#endif
        // the partners' rows of this refresh: asked for right behind the own row
        request_rows(nxt.q0, R + 1);
        if (lane == 0) {
            if constexpr (P2P) {
                peers_lp.store((size_t)row, -0.5 * cur);
            } else {
                log_post[row] = -0.5 * cur;
            }
            zchain[row] = c;
        }
        if (accepted && cur < best) {
            best = cur;
            if (mine)
                bestx[(size_t)lc * nf + lane] = xk;
            if (lane == 0)
                best_iter[lc] = R;
        }
#ifdef MC3HIP_STAMPS
        MC3_TICK(tk_end);
        if (!anyout) {
            ta8 += tk1 - tk0; tr0 += tkP - tk1; tr1 += tkT - tkP; tr2 += tkC - tkT;
            tr3 += tk2 - tkC; ta10 += tk3 - tk2; ta11 += tk_end - tk3; ta12 += 1;
        }
#endif
    }
    if (mine) {
        P.X[(size_t)lc * nf + lane] = xk;
        if (oob)
            atomicAdd(&P.outbounds[lane], oob);
    }
    if (lane == 0) {
        P.chisq[lc] = cur;
        P.best_chisq[lc] = best;
        P.numaccept[lc] = nacc0 + nacc;
    }
#ifdef MC3HIP_STAMPS
    if (lane == 0 && lc == 0) {
        P.stamps[8] += ta8; P.stamps[10] += ta10; P.stamps[11] += ta11;
        P.stamps[12] += ta12; P.stamps[13] += ta13;
        P.stamps[16] += tr0; P.stamps[17] += tr1; P.stamps[18] += tr2; P.stamps[19] += tr3;
    }
#endif
}

// The tiled plan in one launch (ResidentTile above): grid = tiles x columns of chains, eight
// waves; wave v of workgroup w also owns chain w + v * workgroups.  P2P: the ensemble is
// sharded over GPUs, each running this kernel over its own chains; owners publish their Z
// rows into every rank's replica (run_chain P2P), proposals and tile sums stay local.  LDS:
// group totals
// [RS_WCAP][16], the give-up word, a run_chain area per wave, the tile's 2048 weights, and
// (three blocks of operand rows) the last block, [16][512].
// (LEAN: the owners run lean_resident_owner; the two owners are two kernels, so that the one
// of the common shape does not carry the general chain's registers)
template <int KB, bool P2P = false, bool LEAN = false>
__global__ __launch_bounds__(512) void resident_tile_kernel(const DevProblem P, const int nref,
                                                            const int chain_block)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int OWN = 64 + MC3HIP_MAX_PARAMS + 2 + 48 + 4;     // carve() of a one-wave chain
    constexpr int W = RS_WCAP * 16;                              // group totals
    int *s_stop = reinterpret_cast<int *>(smem + W);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    ResidentTile<KB> T;
    T.load(P, blockIdx.x, blockIdx.y, chain_block, smem, s_stop, smem + W + 2 + 8 * OWN);
    // a launch queued behind one that gave up must not run: its inputs are missing
    if (threadIdx.x == 0)
        *s_stop = (__hip_atomic_load(&P.ctl[CTL_ABORT], __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1 : 0;
    block_sync<8>();
    if (*s_stop != 0)
        return;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    const int own_chain = wg + (int)(gridDim.x * gridDim.y) * wv;
    if (own_chain < P.nlocal) {
        T.set_owner(P, own_chain, smem + W + 2 + wv * OWN);
        if constexpr (LEAN)
            lean_resident_owner<MT_LINEAR, P2P>(P, nref, T, own_chain, smem + W + 2 + wv * OWN);
        else
            run_chain<1, MT_LINEAR, false, false, true, -1, false, PH_RESIDENT, P2P, 0, false,
                      ResidentTile<KB>>(P, P.ctl[0], nref, 0, -1, &T, own_chain,
                                        smem + W + 2 + wv * OWN);
    } else {
        // no chain of its own: the tile phase of every iteration, in step with the owners
        for (int R = 0; R < nref; ++R)
            for (int j = 0; j < P.thinning; ++j)
                if (!T.tile_phase(P, true))
                    return;
    }
}

template <int KX, bool P2P = false, bool LEAN = false>
__global__ __launch_bounds__(512) void resident_poly_kernel(const DevProblem P, const int nref,
                                                            const int chain_block)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int OWN = 64 + MC3HIP_MAX_PARAMS + 2 + 48 + 4;     // carve() of a one-wave chain
    constexpr int X = 2 * 16 * 512;                              // the exchange buffers
    int *s_stop = reinterpret_cast<int *>(smem + X);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    ResidentPoly<KX> T;
    T.load(P, blockIdx.x, blockIdx.y, chain_block, smem, s_stop);
    // a launch queued behind one that gave up must not run: its inputs are missing
    if (threadIdx.x == 0)
        *s_stop = (__hip_atomic_load(&P.ctl[CTL_ABORT], __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1 : 0;
    block_sync<8>();
    if (*s_stop != 0)
        return;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    const int own_chain = wg + (int)(gridDim.x * gridDim.y) * wv;
    if (own_chain < P.nlocal) {
        T.set_owner(P, own_chain, smem + X + 2 + wv * OWN);
        if constexpr (LEAN)
            lean_resident_owner<KX, P2P>(P, nref, T, own_chain, smem + X + 2 + wv * OWN);
        else
            run_chain<1, KX, false, false, true, -1, false, PH_RESIDENT, P2P, 0, false,
                      ResidentPoly<KX>>(P, P.ctl[0], nref, 0, -1, &T, own_chain,
                                        smem + X + 2 + wv * OWN);
    } else {
        for (int R = 0; R < nref; ++R)
            for (int j = 0; j < P.thinning; ++j)
                if (!T.tile_phase(P, true))
                    return;
    }
}


// ---- the resident tiled plan with nobody in step with anybody ------------------------------
// resident_tile_kernel above runs an iteration as three phases the whole grid goes through
// together: the owners propose, every workgroup adds up its tile for the chains of its column,
// the owners collect and decide -- and while the owners work (10 500 of C5's 71 000 cycles,
// tools/ticks.py c5) the matrix pipes idle, as they do while a column waits for its slowest
// proposal.  Nothing in the algorithm asks for that: chain c's proposal of iteration R + 1
// needs its own decision of iteration R and those of its TWO partners (chain.py:221-232), not
// the whole ensemble's, and a block of sixteen chains in the matrix instruction is sixteen
// independent columns of the product -- WHICH sixteen chains share a block changes no bit of
// any chain's sum.  So here
//   * the owners are waves of their own (8 .. 8 + RA_OWNERS - 1 of every workgroup, one chain
//     each: lean_resident_owner<.., ASYNC>), not tile waves: a chain is decided as soon as its
//     sums are there, whatever the tile waves are at;
//   * the blocks of an iteration are made up at run time, per column of chains, from the
//     chains whose proposal IS THERE: a scheduler wave (the twelfth wave of the column's
//     workgroup of tile 0) looks at element 0 of the rows, takes up to sixteen chains, lowest
//     first -- never fewer than would add a block to the iteration -- and publishes the block
//     (P.rs_sched); all chains of iteration R before any of R + 1, which keeps the two-row
//     parity scheme of rs_prop and the slots of rs_part what they were (a chain's proposal of
//     R + 1 exists only after every wave of every tile has added its share of R).  Every tile
//     wave of the column takes the blocks in that order, each at its own pace, a block ahead
//     of its products.  (ONE order per column: a chain's sum is complete when the LAST of the
//     column's 49 x 8 waves has taken it -- with every wave making up its own blocks, each
//     chain was in somebody's last block, and the iteration was in step again: 39 us.)
//   * the eight waves' row totals of a chain meet in LDS without a barrier: a wave leaves its
//     total and counts itself in (ds_inc, wrapping at eight); the wave that counts last adds
//     the eight up -- strides 4..1, the order tile_phase's sixteen-lane tree has -- and
//     publishes the tile's sum.
//   * WHICH chains a column of workgroups serves changes from round to round
//     (ResidentColumns): the ensemble's blocks do not divide by the columns as a rule (C5: 32
//     blocks, 5 columns), and the columns that carry one block more take turns at it.
// At the start of an iteration the chains whose partners were decided early in the previous
// one are there (two thirds of them), and by the time those are done so are the others: the
// tile waves only wait when the owners' path is longer than five blocks of matrix work.
// (What an iteration costs, measured by cutting the blocks short -- MC3HIP_RA_FAKE_END, wrong
// sums, right timing -- for C5: 10.3 us + 1.125 us per step of the sixteen of a block, i.e. a
// step of a block costs 0.16 us where the pipe alone would need 0.20 for the pair of waves,
// and ten microseconds per iteration are the chains' own round -- sums, owner, scheduler,
// rows: round trips -- showing through wherever the next blocks are not there yet.  Giving
// the two waves of a SIMD different priorities so that they run half a block apart changes
// nothing: 28.2-28.5 us at 4, 8 and 12 steps apart as without.)
// Same arithmetic in the same order for every chain: the checker cannot tell which of the two
// kernels ran (tests/test_gpu_resident.py).
// (RA_CMAX chains of a column, RA_OWNERS owner waves: core.h)
// the steps of a block's sixteen before which a tile wave looks for the next block (the second
// time if the first look found nothing)
#ifndef MC3HIP_RA_TAKE_AT
#define MC3HIP_RA_TAKE_AT 6
#endif
#ifndef MC3HIP_RA_TAKE_AGAIN
#define MC3HIP_RA_TAKE_AGAIN 12
#endif
constexpr int RA_TAKE_AT = MC3HIP_RA_TAKE_AT, RA_TAKE_AGAIN = MC3HIP_RA_TAKE_AGAIN;
#ifndef MC3HIP_RA_PRIO
#define MC3HIP_RA_PRIO 2
#endif

__device__ __forceinline__ unsigned lds_offset(const void *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}

// A wave's row totals of the four chains of a lane (slots g, g + 4, g + 8, g + 12 of its block)
// into their slots, and the wave counted in for each: old[r] = 7 = it is the last of the eight.
// One inline assembly -- a counter must not overtake its total, no fence the compiler knows
// would let the proposal rows already asked for stay in flight, and four LDS round trips one
// after the other (with the adding-up of a chain between them) were a third of the time
// between two blocks.
__device__ __forceinline__ void tile_part_arrive4(double *const (&slot)[4], const double (&v)[4],
                                                  unsigned *const (&cnt)[4], unsigned (&old)[4])
{
    asm volatile("ds_write_b64 %4, %8\n\tds_inc_rtn_u32 %0, %12, %16\n\t"
                 "ds_write_b64 %5, %9\n\tds_inc_rtn_u32 %1, %13, %16\n\t"
                 "ds_write_b64 %6, %10\n\tds_inc_rtn_u32 %2, %14, %16\n\t"
                 "ds_write_b64 %7, %11\n\tds_inc_rtn_u32 %3, %15, %16\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3])
                 : "v"(lds_offset(slot[0])), "v"(lds_offset(slot[1])), "v"(lds_offset(slot[2])),
                   "v"(lds_offset(slot[3])), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]),
                   "v"(lds_offset(cnt[0])), "v"(lds_offset(cnt[1])), "v"(lds_offset(cnt[2])),
                   "v"(lds_offset(cnt[3])), "v"(7u)
                 : "memory");
}

// The same for ONE chain per lane (row_totals4 below leaves the four chains of a sixteen-lane
// row in its four quads): a quarter of the LDS operations per wave, one chain to finish per
// lane instead of four one after the other.
__device__ __forceinline__ unsigned tile_part_arrive1(double *slot, const double v, unsigned *cnt)
{
    unsigned old;
    asm volatile("ds_write_b64 %1, %2\n\tds_inc_rtn_u32 %0, %3, %4\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(old)
                 : "v"(lds_offset(slot)), "v"(v), "v"(lds_offset(cnt)), "v"(7u)
                 : "memory");
    return old;
}

// lane l of a sixteen-lane row receives lane l ^ 4's value (two row shifts, each written to
// the banks it is valid for)
__device__ __forceinline__ double row_xor4(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x104, 0xf, 0x5, false);      // row_shl:4
    int hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x104, 0xf, 0x5, false);
    lo2 = __builtin_amdgcn_update_dpp(lo2, lo, 0x114, 0xf, 0xa, false);        // row_shr:4
    hi2 = __builtin_amdgcn_update_dpp(hi2, hi, 0x114, 0xf, 0xa, false);
    return make_double(lo2, hi2);
}
template <int CTRL>
__device__ __forceinline__ double add_quad_perm(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    const int hi2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return v + make_double(lo2, hi2);
}

// The sixteen-lane totals of the four accumulators of a lane (chains r = 0..3 of its row),
// transposed: five additions instead of sixteen on a pipe the wave shares with its partner's
// matrix instructions.  Stride 8 pairs the lanes of TWO chains per addition (a lane keeps the
// chains of its half of the row and hands over the other two), stride 4 those of the two it
// kept, strides 2 and 1 run inside a quad: lane j ends with the total of chain j >> 2 -- the
// pairs (j, j + 8), (.., + 4), (.., + 2), (.., + 1) of four add_ror trees, level for level, so
// every total is bit for bit theirs (an addition's operands change sides, which changes no
// bit of a sum).
__device__ __forceinline__ double row_totals4(const double (&acc)[4], const int j)
{
    const bool hi8 = (j & 8) != 0, hi4 = (j & 4) != 0;
    const double k0 = hi8 ? acc[2] : acc[0], s0 = hi8 ? acc[0] : acc[2];
    const double k1 = hi8 ? acc[3] : acc[1], s1 = hi8 ? acc[1] : acc[3];
    const int s0l = __builtin_amdgcn_mov_dpp(__double2loint(s0), 0x128, 0xf, 0xf, true);
    const int s0h = __builtin_amdgcn_mov_dpp(__double2hiint(s0), 0x128, 0xf, 0xf, true);
    const int s1l = __builtin_amdgcn_mov_dpp(__double2loint(s1), 0x128, 0xf, 0xf, true);
    const int s1h = __builtin_amdgcn_mov_dpp(__double2hiint(s1), 0x128, 0xf, 0xf, true);
    const double t0 = k0 + make_double(s0l, s0h);      // chain 0 (lanes 0..7) / 2 (lanes 8..15)
    const double t1 = k1 + make_double(s1l, s1h);      // chain 1 / 3
    const double k = hi4 ? t1 : t0, s = hi4 ? t0 : t1;
    double u = k + row_xor4(s);
    u = add_quad_perm<0x4e>(u);                        // quad_perm [2, 3, 0, 1]
    return add_quad_perm<0xb1>(u);                     // quad_perm [1, 0, 3, 2]
}

__device__ __forceinline__ unsigned long long load_sched(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// false: the launch is being given up (said to everybody if it was this wave that lost patience)
__device__ __forceinline__ bool async_patience(unsigned &spins, long long *ctl)
{
    if (__builtin_expect((++spins & 255u) == 0u, 0)) {
        const int ab = __builtin_amdgcn_readfirstlane(
            (int)(__hip_atomic_load(&ctl[CTL_ABORT], __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT) != 0));
        if (ab != 0 || spins > (1u << 22)) {
            if (ab == 0)
                __hip_atomic_store(&ctl[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

// Which chains a column of workgroups serves, round by round.  The ensemble is nb = ceil(nlocal /
// 16) blocks of sixteen chains; ncol columns share them as evenly as whole blocks allow -- base
// = nb / ncol each, and one more for nb % ncol of them -- and WHICH columns carry the extra
// block moves on by one column per round: over ncol rounds every workgroup has multiplied
// the same number of blocks, nb / ncol per round (C5: 6.4, where columns of fixed height do 7, 7,
// 7, 7, 4 and the chip waits for the sevens).  The owners know nothing of columns (a chain's
// proposal row and its tile sums are found by the chain's number), and the scheduler and the
// tile waves of a column count the launch's rounds the same way.
struct ResidentColumns {
    int base, rem, ncol, nlocal, col, rot;
    __device__ __forceinline__ void init(const DevProblem &P, const int column, const int ncolumns)
    {
        const int nb = (P.nlocal + 15) / 16;
        ncol = ncolumns;
        base = nb / ncol;
        rem = nb - base * ncol;
        nlocal = P.nlocal;
        col = column;
        rot = 0;
    }
    // the chains of the column in the current round: c0 .., n of them (n >= 1)
    __device__ __forceinline__ void chains(int &c0, int &n) const
    {
        // columns k with (k + rot) mod ncol < rem carry base + 1 blocks: how many before col
        const int e = rot + col;
        const int e1 = e < ncol ? e : ncol;
        int extra = (e1 < rem ? e1 : rem) - rot;
        extra = extra > 0 ? extra : 0;
        if (e > ncol)
            extra += (e - ncol < rem) ? e - ncol : rem;
        const int pos = e < ncol ? e : e - ncol;
        const int b0 = col * base + extra, nb = base + (pos < rem ? 1 : 0);
        c0 = 16 * b0;
        const int c1 = 16 * (b0 + nb);
        n = (c1 < nlocal ? c1 : nlocal) - c0;
    }
    __device__ __forceinline__ void next_round() { rot = (rot + 1 == ncol) ? 0 : rot + 1; }
};

// The scheduler wave of a column: the blocks of every iteration of the launch, in
// P.rs_sched's rows of the column.  sel: 16 ints of LDS of its own.
__device__ __forceinline__ void resident_scheduler(const DevProblem &P, const int nref,
                                                   const int col, const int ncol, int *sel)
{
    long long *ctl = const_cast<long long *>(P.ctl);
    const int lane = threadIdx.x & 63;
    unsigned long long *rows = P.rs_sched + (size_t)col * P.rs_sched_rows * 16;
    const unsigned long long gen = (unsigned long long)(unsigned)P.rs_gen << 32;
    unsigned spins = 0;
    long long seq = 0;
    ResidentColumns C;
    C.init(P, col, ncol);
    for (int R = 0; R < nref; ++R, C.next_round()) {
        int c0, n;
        C.chains(c0, n);
        const int l0 = (lane < n) ? lane : 0, l1 = (lane + 64 < n) ? lane + 64 : 0;
        // (the word a chain's owner leaves behind its proposal of this round: rs_tag)
        const unsigned long long want = gen | (unsigned long long)(unsigned)(R + 1);
        // chains lane and lane + 64 of the column: still to do in this iteration
        bool rem0 = lane < n, rem1 = lane + 64 < n;
        bool r0 = false, r1 = false;        // ... and known to be there
        bool poll = true;
        int nrem = n;
        while (nrem > 0) {
            // is the proposal of THIS round there?  (the owner's word for the chain; a tile wave
            // waits for the row's elements when it uses the row)  After a block: the next one
            // at once, without asking again, if the chains already known to be there make one up.
            if (poll) {
                const unsigned long long f0 = load_sched(P.rs_tag + c0 + l0);
                const unsigned long long f1 = (n > 64) ? load_sched(P.rs_tag + c0 + l1) : 0ull;
                r0 = rem0 && f0 == want;
                r1 = rem1 && f1 == want;
            }
            poll = true;
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(r0);
            const unsigned long long m1 = __builtin_amdgcn_ballot_w64(r1);
            const int p0 = __builtin_popcountll(m0), p1 = __builtin_popcountll(m1);
            const int nready = p0 + p1;
            const int ragged = nrem - 16 * ((nrem + 15) / 16 - 1);     // 1..16
            if (nready < ragged) {
                if (!async_patience(spins, ctl))
                    return;
                continue;
            }
            spins = 0;
            if (seq >= (long long)P.rs_sched_rows) {
                // (never with the rows the host sizes: a row per block of a launch's rounds)
                __hip_atomic_store(&ctl[CTL_ABORT], 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            const int rank0 = (int)__builtin_amdgcn_mbcnt_hi(
                (unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u));
            const int rank1 = (p0 < 16 ? p0 : 16) + (int)__builtin_amdgcn_mbcnt_hi(
                (unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u));
            const bool t0 = r0 && rank0 < 16, t1 = r1 && rank1 < 16;
            const int nt = nready < 16 ? nready : 16;
            __builtin_amdgcn_wave_barrier();
            if (t0)
                sel[rank0] = lane;
            if (t1)
                sel[rank1] = lane + 64;
            __builtin_amdgcn_wave_barrier();
            // (slots beyond the block: its first chain again, multiplied and not used)
            const int id = sel[(lane & 15) < nt ? (lane & 15) : 0];
            __builtin_amdgcn_wave_barrier();
            if (lane < 16)
                __hip_atomic_store(&rows[seq * 16 + lane],
                                   gen | ((unsigned long long)nt << 8) | (unsigned long long)id,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef MC3HIP_TIMELINE
            if (R == MC3HIP_TIMELINE) {
                const int b_ = (n - nrem) / 16;
                MC3_TL(lane == 0 && b_ < 16, 4096 + 16 * col + b_);
                if (lane < nt)
                    P.stamps[16384 + c0 + id] = (unsigned long long)b_;
            }
#endif
            ++seq;
            rem0 = rem0 && !t0;
            rem1 = rem1 && !t1;
            r0 = r0 && !t0;
            r1 = r1 && !t1;
            nrem -= nt;
            poll = false;
        }
    }
}

// (diagnostic build: the tile wave's phases on the 100 MHz clock -- the shader clock's own
// count stalls while FP64 matrix instructions run: 5 018 "cycles" for a pair of blocks that
// takes 3.18 us, tools/probes/tile_products_probe.hip)
#ifdef MC3HIP_STAMPS
#define MC3_RTICK(var)                                                                    \
    do {                                                                                  \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");    \
    } while (0)
#else
#define MC3_RTICK(var) do { } while (0)
#endif

// (two blocks of operand rows in registers, the others in LDS: with twelve waves on the CU a
// wave has 168 registers, and three blocks -- 96 of them -- leave too few for the rest)
// (KRMAX = 1 beside the general chain as owner: its scalar state, spilled to vector lanes, takes
// registers from every wave of the kernel -- an operand of the tile was reloaded from scratch at
// the head of every block, a wait for everything the wave has in flight)
template <int KB, int KRMAX = 2>
struct ResidentTileAsync : ResidentTile<KB, KRMAX> {
    static constexpr int KR = ResidentTile<KB, KRMAX>::KR;
    static constexpr int KL = ResidentTile<KB, KRMAX>::KL;

    // a tile wave, for the whole launch.  s_part [2][RA_CMAX + 1][8]: the waves' row totals of
    // a chain of the column (by the round's parity: ResidentColumns); s_cnt [2][RA_CMAX + 1]:
    // waves counted in; s_sel [9][2][16]: the chains of a wave's blocks
    // (the one in hand, the next); s_prog [8]: blocks taken so far, by wave.
    // A turn of the loop multiplies the block in hand and, meanwhile, takes the one after: it
    // looks at the scheduler's next row twice -- RA_TAKE_AT steps into the products (the row
    // was asked for at the end of the turn before) and, if it was not there, again at step
    // RA_TAKE_AGAIN -- and asks for the rows of the chains it finds, which travel behind the
    // rest of the products.  A turn with no block in hand (the launch's first, or both looks of
    // the turn before found nothing) only looks.
    // Where loads go is laid out by hand: two places that load into the SAME variables meet at
    // a join where the compiler copies one into the other -- and waits for the load to copy
    // it, i.e. before the products the load should travel behind.  So the next row is asked
    // for in ONE place (the end of a turn), the second look has variables of its own (w2, the
    // rows of its chains in aop2), and what was loaded is copied only where it is waited for
    // anyway.  The sums are published AFTER the wait for the next block's rows: the wait is
    // for everything the wave has in flight (vmcnt(0)), and behind a store that is a round trip
    // for its acknowledgement.
    // s_done [8] (behind s_prog): blocks whose row totals a wave has left in the slots so far.
    // The slots of a chain are keyed by its place in the column and the round's parity, and
    // the columns' chains change from round to round: a wave enters round R only when all
    // eight have left R - 2's slots (nothing else keeps two waves of a workgroup less than two
    // rounds apart) -- it looks at the scheduler's row again instead, as if nothing were there.
    __device__ __forceinline__ void run(const DevProblem &P, const int nref, double *s_part,
                                        unsigned *s_cnt, int *s_sel, int *s_prog)
    {
        int *const s_done = s_prog + 8;
        int fin = 0;                        // blocks whose totals this wave has left
        int gate_old = 0, gate_new = 0;     // blocks up to the end of rounds R - 2, R - 1
        long long *ctl = const_cast<long long *>(P.ctl);
        const int lane = this->lane, j = lane & 15, g = lane >> 4;
        const int wv = this->wv, t = this->t;
        ResidentColumns C;
        C.init(P, blockIdx.y, gridDim.y);
        int c0, n;                          // the column's chains in round R
        C.chains(c0, n);
        int *const sel2 = s_sel + 32 * wv;
        const unsigned long long *rows =
            P.rs_sched + (size_t)blockIdx.y * P.rs_sched_rows * 16 + j;
        const unsigned gen = (unsigned)P.rs_gen;
        int R = 0, nrem = n;
        bool done = false;                  // every block of the launch has been taken
        bool have = false;                  // a block is in hand: a[], cur_nt, cur_sel
        long long seq = 0;                  // the scheduler's next block
        unsigned long long w = load_sched(rows), w2 = 0;
        double aop1[KB], aop2[KB], a[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
            aop1[kb] = aop2[kb] = a[kb] = 0.0;
        int nt = 0, lc_next = c0, buf = 0, cur_nt = 0;
        // (of the block taken / in hand: the first chain of its round's column, and which of
        // the two sets of LDS slots -- by the round's parity -- its sums meet in)
        int next_c0 = c0, next_slots = 0, cur_c0 = c0, cur_slots = 0;
        const int *cur_sel = sel2;
        int wo = 0;
        unsigned spins = 0;
#ifdef MC3HIP_TIMELINE
        int tl_next = -1, tl_cur = -1;
#endif
#ifdef MC3HIP_STAMPS
        // (diagnostic build: wave 0 of workgroup 0 -- blocks multiplied, turns with nothing in
        // hand, cycles multiplying, waiting for rows, in all)
        unsigned long long tq_blocks = 0, tq_idle = 0, tq_mul = 0, tq_wait = 0, tq_begin = 0;
        unsigned long long tq_arrive = 0, tq_finish = 0, tq_turn = 0, tq_last = 0;
        MC3_RTICK(tq_begin);
#endif
                // the scheduler's next block, if word ww says it is there (word j = slot j's chain)
        auto take = [&](const unsigned long long ww) -> bool {
            if (!__all((unsigned)(ww >> 32) == gen))
                return false;
            if (nrem == n && R >= 2) {      // (a round's first block)
                const int d = __hip_atomic_load(&s_done[lane & 7], __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!__all(d >= gate_old))
                    return false;
            }
            nt = __builtin_amdgcn_readfirstlane((int)((unsigned)ww >> 8) & 0xff);
            const int id = (int)((unsigned)ww & 0xffu);
            int *const sel = sel2 + 16 * buf;
            __builtin_amdgcn_wave_barrier();
            if (lane < 16)
                sel[lane] = id;
            __builtin_amdgcn_wave_barrier();
            lc_next = (R & 1) * P.nlocal + c0 + id;
            next_c0 = c0;
            next_slots = (R & 1) * (RA_CMAX + 1);
            buf ^= 1;
            // The two tile waves of a SIMD (v and v ^ 4) share its FP64 pipe, and the arbiter
            // serves the older one first: left alone, waves 0..3 run blocks ahead of 4..7 --
            // and a chain's sum is complete when the LAST wave has added its share
            // (tools/timeline_c5.py: sums collected 10 us after wave 0 had left the block).
            // The one that is behind goes first.
            {
                s_prog[wv] = (int)seq;
                const int other = s_prog[wv ^ 4];
                if ((int)seq > other)
                    __builtin_amdgcn_s_setprio(0);
                else
                    __builtin_amdgcn_s_setprio(1);
            }
#ifdef MC3HIP_TIMELINE
            tl_next = (R == MC3HIP_TIMELINE && wv == 0) ? (n - nrem) / 16 : -1;
            MC3_TL(tl_next >= 0 && tl_next < 8 && lane == 0,
                   8192 + 32 * (blockIdx.y * gridDim.x + blockIdx.x) + 4 * tl_next);
#endif
            ++seq;
            nrem -= nt;
            if (nrem == 0) {
                gate_old = gate_new;
                gate_new = (int)seq;
                if (R + 1 == nref) {
                    done = true;
                } else {
                    ++R;
                    C.next_round();
                    C.chains(c0, n);
                    nrem = n;
                }
            }
            return true;
        };
        for (;;) {
            unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0;
            (void)tq0; (void)tq1; (void)tq2; (void)tq3; (void)tq4;
            double acc[4] = {0.0, 0.0, 0.0, 0.0};      // chain of slot g + 4 r: this lane's points
            // Steps S0 .. S1 - 1 of the block in hand, in the order they are to be issued: the
            // products of step s + 1 BEFORE the four multiply-adds of step s (whose products
            // have long left the pipe by then: no wait), the LDS operands two steps ahead --
            // one wave alone keeps the FP64 pipe busy that way, whatever its partner on the
            // SIMD is doing.  (tools/probes/tile_products_probe.hip: 3.19 us per block and
            // pair of waves, 3.65 in the compiler's order.)
            constexpr int KLX = KL > 0 ? KL : 1;
            auto lds_ops = [&](const int st, double(&bl)[KLX]) {
#pragma unroll
                for (int q = 0; q < KL; ++q)
                    bl[q] = this->s_b[wo + (q * 16 + st) * 512];
            };
            auto chain = [&](const int st, const double(&bl)[KLX]) {
                const int vb = st >> 1, h = st & 1;
                mc3_v4d m = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    m = __builtin_amdgcn_mfma_f64_16x16x4f64(
                        a[kb],
                        (kb < KR) ? this->bop[vb][h][kb < KR ? kb : 0] : bl[kb < KR ? 0 : kb - KR],
                        m, 0, 0, 0);
                return m;
            };
            auto products = [&](const int S0, const int S1) {
                if (S0 >= S1)
                    return;
                double bl[3][KLX];
                lds_ops(S0, bl[S0 % 3]);
                if (S0 + 1 < S1)
                    lds_ops(S0 + 1, bl[(S0 + 1) % 3]);
                mc3_v4d m[2];
                m[S0 & 1] = chain(S0, bl[S0 % 3]);
#pragma unroll
                for (int st = S0; st < S1; ++st) {
                    if (st + 1 < S1) {
                        if (st + 2 < S1)
                            lds_ops(st + 2, bl[(st + 2) % 3]);
                        m[(st + 1) & 1] = chain(st + 1, bl[(st + 1) % 3]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[r] = mc3_fma(m[st & 1][r], m[st & 1][r], acc[r]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            static_assert(RA_TAKE_AT < RA_TAKE_AGAIN && RA_TAKE_AGAIN < 16, "");
            if (have && P.test_slow_ticks > 0 && (int)blockIdx.y == P.test_slow_col) {
                // test hook: this column's tile waves are slow (core.h test_slow_col)
                unsigned long long t0, t1;
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
                do {
                    __builtin_amdgcn_s_sleep(8);
                    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
                } while (t1 - t0 < (unsigned long long)P.test_slow_ticks);
            }
            if (have) {
                MC3_RTICK(tq0);
#ifdef MC3HIP_TIMELINE
                MC3_TL(tl_cur >= 0 && tl_cur < 8 && lane == 0,
                       8192 + 32 * (blockIdx.y * gridDim.x + blockIdx.x) + 4 * tl_cur + 2);
#endif
                asm volatile("; operands of this block" : "+v"(wo));
                products(0, RA_TAKE_AT);
            }
            // the first look
            bool hit1 = false, hit2 = false;
            int lc1 = lc_next;
            const bool look = !done;
            if (look) {
                hit1 = take(w);
                lc1 = lc_next;
                // the rows of the block's chains (nothing there: the last one's again, unused)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    aop1[kb] = load_published(P.rs_prop + (size_t)lc1 * RS_ROW + 4 * kb + g);
                if (!hit1)
                    w2 = load_sched(rows + seq * 16);
            }
            if (have)
                products(RA_TAKE_AT, RA_TAKE_AGAIN);
            // the second look
            if (look && !hit1) {
                hit2 = take(w2);
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    aop2[kb] = load_published(P.rs_prop + (size_t)lc_next * RS_ROW + 4 * kb + g);
            }
            if (have) {
                products(RA_TAKE_AGAIN, 16);
            }
            const bool ok = hit1 || hit2;
            MC3_RTICK(tq2);
            // the rows of the block taken: there by now (only a row that is late is asked for
            // a second time: the addresses again, from the row's number)
            if (hit1) {
                const double *addr[KB];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    addr[kb] = P.rs_prop + (size_t)lc1 * RS_ROW + 4 * kb + g;
                if (!wait_rows<KB>(addr, aop1, ctl))
                    return;
            } else if (hit2) {
                const double *addr[KB];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    addr[kb] = P.rs_prop + (size_t)lc_next * RS_ROW + 4 * kb + g;
                if (!wait_rows<KB>(addr, aop2, ctl))
                    return;
            }
            MC3_RTICK(tq3);
            if (have) {
                // The sums of the block multiplied -- AFTER the wait above: that wait is for
                // everything the wave has in flight, and behind a store it is a round trip
                // for the store's acknowledgement.  (Left for later still -- behind the next
                // block's first products, where they would cost the pipe nothing -- they reach
                // the owners two microseconds later, and the tile waves run out of proposals:
                // 33.2 us per iteration against 29.5.)
                // the 16 lanes of a row, the four chains of a lane together: lane j has the
                // total of slot g + 4 (j >> 2) (strides 8..1, the order add_ror's tree has)
                const double sr = row_totals4(acc, j);
                if ((j & 3) == 0) {
                    // (a slot beyond the block's chains goes to a row of LDS that nobody reads)
                    const int sl = g + (j & 12);
                    const int ch = (sl < cur_nt) ? cur_sel[sl] : RA_CMAX;
                    const unsigned old = tile_part_arrive1(&s_part[(cur_slots + ch) * 8 + wv], sr,
                                                           &s_cnt[cur_slots + ch]);
                    MC3_RTICK(tq4);
                    // the chain this wave counted last for: the tile's eight rows (strides 4..1)
                    if (old == 7u) {
                        const double *p = s_part + (cur_slots + ch) * 8;
                        double sg = ((p[0] + p[4]) + (p[2] + p[6])) +
                                    ((p[1] + p[5]) + (p[3] + p[7]));
                        if (mc3_bits(sg) == ROW_PENDING)   // (a NaN that looks like "not there yet")
                            sg = mc3_from_bits(0x7ff8000000000000ull);
                        if (ch != RA_CMAX)
                            __hip_atomic_store(&P.rs_part[(size_t)(cur_c0 + ch) * P.rs_ntp + t], sg,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                // (every lane writes the one word, from inline assembly: masking the store down
                // to a lane, or leaving it to the compiler to place, cost 0.1-0.7 us per
                // iteration of C5 -- A/B on one box, docs/HISTORY.md round 6)
                ++fin;
                asm volatile("ds_write_b32 %0, %1"
                             :: "v"(lds_offset(&s_done[wv])), "v"(fin) : "memory");
                __builtin_amdgcn_s_setprio(1);
                spins = 0;
                MC3_RTICK(tq1);
#ifdef MC3HIP_TIMELINE
                MC3_TL(tl_cur >= 0 && tl_cur < 8 && lane == 0,
                       8192 + 32 * (blockIdx.y * gridDim.x + blockIdx.x) + 4 * tl_cur + 3);
#endif
#ifdef MC3HIP_STAMPS
                tq_blocks += 1;
                tq_mul += tq2 - tq0;
                tq_wait += tq3 - tq2;
                if (j == 0) {
                    tq_arrive += tq4 - tq3;
                    tq_finish += tq1 - tq4;
                }
                if (tq_last != 0)
                    tq_turn += tq0 - tq_last;
                tq_last = tq1;
#endif
            } else if (!ok) {
                if (done)
                    break;                              // the launch's last block is behind us
                // nothing in hand, nothing there: look again -- not for ever
#ifdef MC3HIP_STAMPS
                tq_idle += 1;
#endif
                if (!async_patience(spins, ctl))
                    return;
            }
            have = ok;
#ifdef MC3HIP_TIMELINE
            tl_cur = ok ? tl_next : -1;
            MC3_TL(tl_cur >= 0 && tl_cur < 8 && lane == 0,
                   8192 + 32 * (blockIdx.y * gridDim.x + blockIdx.x) + 4 * tl_cur + 1);
#endif
            if (ok) {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    a[kb] = hit1 ? aop1[kb] : aop2[kb];
                cur_nt = nt;
                cur_sel = sel2 + 16 * (buf ^ 1);
                cur_c0 = next_c0;
                cur_slots = next_slots;
            }
            // the scheduler's row after the block taken (or the same one again)
            if (!done)
                w = load_sched(rows + seq * 16);
        }
#ifdef MC3HIP_STAMPS
        if (lane == 0) {
            // every tile wave: ticks waiting for rows, turns with nothing in hand, ticks in the
            // products, in all (tools/tile_wave_spread.py)
            unsigned long long tq_end = 0;
            MC3_RTICK(tq_end);
            unsigned long long *st =
                P.stamps + 4096 + 4 * ((blockIdx.y * gridDim.x + blockIdx.x) * 8 + wv);
            st[0] += tq_wait; st[1] += tq_idle; st[2] += tq_mul; st[3] += tq_end - tq_begin;
        }
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
            unsigned long long tq_end = 0;
            MC3_RTICK(tq_end);
            P.stamps[20] += tq_blocks; P.stamps[21] += tq_idle; P.stamps[22] += tq_mul;
            P.stamps[23] += tq_wait; P.stamps[24] += tq_end - tq_begin;
            P.stamps[25] += tq_arrive; P.stamps[26] += tq_finish; P.stamps[27] += tq_turn;
        }
#endif
    }
};

// grid = tiles x columns of chains as resident_tile_kernel's; 8 tile waves, up to RA_OWNERS
// owner waves (wave 8 + v of workgroup w owns chain w + v * workgroups), and in the workgroups
// of tile 0 the column's scheduler behind them.  LDS: the waves' row totals [RA_CMAX][8],
// their counters, the blocks' chains, the give-up word, an area per owner, and the blocks of
// operand rows beyond two, [16][512] each.
// (LEAN: the owners run lean_resident_owner -- DEMC, thinning 1, no trace, no shared parameters;
// otherwise the general run_chain: any sampler, thinning, shared parameters)
template <int KB, bool P2P = false, bool LEAN = true>
__global__ __launch_bounds__(512 + 64 * (RA_OWNERS + 1)) void resident_async_kernel(
    const DevProblem P, const int nref, const int chain_block)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int OWN = 64 + MC3HIP_MAX_PARAMS + 2 + 48 + 4;     // carve() of a one-wave chain
    double *s_part = smem;                              // (core.h RA_HEAD)
    unsigned *s_cnt = reinterpret_cast<unsigned *>(smem + RA_PART);
    int *s_sel = reinterpret_cast<int *>(smem + RA_PART + RA_CNT);
    int *s_stop = reinterpret_cast<int *>(smem + RA_PART + RA_CNT + 144);
    double *s_own = smem + RA_HEAD;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    ResidentTileAsync<KB, (LEAN || KB == 4) ? 2 : 1> T;
    if (wv < 8)
        T.load(P, blockIdx.x, blockIdx.y, chain_block, smem, s_stop, s_own + RA_OWNERS * OWN);
    if (threadIdx.x < 2 * RA_CNT)
        s_cnt[threadIdx.x] = 0u;
    if (threadIdx.x < 16)
        s_sel[32 * 8 + 16 + threadIdx.x] = 0;           // (the tile waves' progress: s_prog, s_done)
    // a launch queued behind one that gave up must not run: its inputs are missing
    if (threadIdx.x == 0)
        *s_stop = (__hip_atomic_load(&P.ctl[CTL_ABORT], __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1 : 0;
    __syncthreads();
    if (*s_stop != 0)
        return;
    // (rounds of proposals of the launch: thinning iterations to a refresh)
    const int rounds = nref * P.thinning;
    if (wv < 8) {
        T.run(P, rounds, s_part, s_cnt, s_sel, s_sel + 32 * 8 + 16);
        return;
    }
    // (owners and scheduler: few instructions, and every one of them on the path between a
    // block's sums and the next proposals -- ahead of the tile waves' products)
    __builtin_amdgcn_s_setprio(MC3HIP_RA_PRIO);
    if (wv == nwaves - 1) {
        if (blockIdx.x == 0) {
            resident_scheduler(P, rounds, blockIdx.y, gridDim.y, s_sel + 32 * 8);
        }
        return;
    }
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    const int own_chain = wg + (int)(gridDim.x * gridDim.y) * (wv - 8);
    if (own_chain < P.nlocal) {
        ResidentOwnerAsync O;
        O.lane = threadIdx.x & 63;
        O.par = 0;
        O.failed = false;
        O.tag = (unsigned long long)(unsigned)P.rs_gen << 32;
        O.set_owner(P, own_chain, s_own + (wv - 8) * OWN);
        if constexpr (LEAN)
            lean_resident_owner<MT_LINEAR, P2P, ResidentOwnerAsync, true>(
                P, nref, O, own_chain, s_own + (wv - 8) * OWN);
        else
            run_chain<1, MT_LINEAR, false, false, true, -1, false, PH_RESIDENT, P2P, 0, false,
                      ResidentOwnerAsync>(P, P.ctl[0], nref, 0, -1, &O, own_chain,
                                          s_own + (wv - 8) * OWN);
    }
}

}  // namespace mc3hip
