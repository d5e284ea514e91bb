// resident.h -- the tiled plan as ONE persistent launch.
#pragma once
#include "core.h"
#include "handoff.h"
#include "run_chain.h"

namespace mc3hip {

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

    // owner, after the tile phase: the data sum of chain lc, tiles added in order; hands the
    // slots and the row this iteration used back
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
            // (in order, eight broadcasts at a time ahead of their dependent additions)
            int k = 0;
            for (; k + 8 <= n; k += 8) {
                double v8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v8[u] = read_lane(val[0], k + u);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    total = total + v8[u];
            }
            for (; k < n; ++k)
                total = total + read_lane(val[0], k);
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

template <int KB>
struct ResidentTile : ResidentOwner {
    // operand rows of the tile: blocks of four rows in registers (up to three: 96 registers);
    // a fourth block (12..15 coefficients + the data) lives in LDS, 64 KB, one read per product
    static constexpr int KR = (KB == 4) ? 3 : KB;
    double bop[8][2][KR];
    const double *s_b;      // LDS [16][512]: this thread's operands of row block KR
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
                        bop[vb][h][kb] = v;
                    else
                        lds_b[(2 * vb + h) * 512 + threadIdx.x] = v;
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
template <int MT, bool P2P, class H>
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
        const double jump_k = gamma * (rv[0] - rv[1]) + fepsilon * normal_k;
        // proposal and boundaries, chain.py:235-243
        const double npk = xk + jump_k;
        const bool outk = mine && (npk < pmin_k || npk > pmax_k);
        const bool anyout = __any(outk);
        if (mine)
            s_nextp[jfree] = npk;
        T.publish(P, lc, s_nextp);
        MC3_TICK(tkP);
        alive = T.tile_phase(P, alive && !give_up);
        MC3_TICK(tkT);
        if (!alive)
            break;
        // (also of a proposal that is out of bounds: the sums must be seen before the row is
        // reused)
        const double total = T.collect(P, lc);
        MC3_TICK(tkC);
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

}  // namespace mc3hip
