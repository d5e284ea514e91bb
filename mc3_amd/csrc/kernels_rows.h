// kernels_rows.h -- the hand-laid-out loops for ensembles with more chains than waves can be
// co-resident, the chains of a wave SIDE BY SIDE in its rows of 16 lanes.
//
// Round 3's loops (lean_*_mc_kernel; of them kernels_lean.h keeps the two-tile DEMC one) advanced
// the T = 2..4 chains of a wave one after another.  Measured on MI355X at 8192 chains
// (profiles/r04_c3x4_pmc_before.json) that loop spends 201
// vector instructions per chain-step of which 109 are FP64 arithmetic: the rest are T copies
// of the scalar work of a turn and the register shuffling of T turns' control flow.  Here, for
// one-tile polynomials of 2..4 coefficients, lane 16 r + k holds coefficient k (FX shapes: free
// parameter k) of the chain of row r (T = 4: chain r;
// T = 3: rows 2 and 3 both chain 2; T = 2: rows 0-1 chain 0, rows 2-3 chain 1; a duplicate row
// computes what its twin computes and stores nothing).  Everything but the sixteen residuals
// per lane and chain then runs ONCE per refresh for all the chains of the wave:
//   * one load per partner / history row brings it for all T chains (lane 16 r + k asks for
//     coefficient k of the row the chain of row r needs), one store publishes the T new rows;
//   * proposal, boundaries, Metropolis test, best-so-far: one instruction stream, the chains
//     in different rows; the draws of a batch of refreshes sit one (refresh, slot, chain) per
//     lane and reach their rows through the LDS crossbar (ds_bpermute, no LDS memory);
//   * the T data sums are reduced across lanes TOGETHER (core.h rows_sum: the stride-32 and
//     -16 levels pair two and four chains per instruction, the last four levels serve every
//     row at once) with the pairing of wave_sum, so each total is bit for bit the one a chain
//     alone in its wave computes.
// 8192 chains, DEMC: 123 vector instructions per chain-step (99 FP64), 1.85 -> 2.56 G
// chain-steps/s (profiles/r04_c3x4_pmc.json).  The hand-off protocol, the draws (Philox keyed
// by seed, refresh, chain, slot) and the arithmetic are unchanged: same results, bit for bit.
#pragma once
#include "run_chain.h"

namespace mc3hip {

// a 64-bit field of another lane (byte address 4 * lane) through the crossbar
__device__ __forceinline__ unsigned long long gather_u64(unsigned long long v, int at)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(at, (int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(at, (int)(unsigned)(v >> 32));
    return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
__device__ __forceinline__ double gather_f64(double v, int at)
{
    return make_double(__builtin_amdgcn_ds_bpermute(at, __double2loint(v)),
                       __builtin_amdgcn_ds_bpermute(at, __double2hiint(v)));
}

// lane K of every row of 16 lanes, to all the lanes of its row (DPP row_newbcast)
template <int K>
__device__ __forceinline__ double row_bcast(double v)
{
    return make_double(__builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + K, 0xf, 0xf, true),
                       __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + K, 0xf, 0xf, true));
}
// lane_serial_sum for every row at once: lanes 0..N-1 of the row, in lane order
template <int N>
__device__ __forceinline__ double row_serial_sum(double t)
{
    double s = 0.0;
    s = s + row_bcast<0>(t);
    if constexpr (N > 1)
        s = s + row_bcast<1>(t);
    if constexpr (N > 2)
        s = s + row_bcast<2>(t);
    if constexpr (N > 3)
        s = s + row_bcast<3>(t);
    static_assert(N >= 1 && N <= 4, "up to four coefficients");
    return s;
}

// ... and of lanes 0..n-1, n <= 4 known at run time (FX shapes: n free parameters, n priors)
__device__ __forceinline__ double row_serial_sum_n(double t, int n)
{
    double s = 0.0;
    if (n > 0)
        s = s + row_bcast<0>(t);
    if (n > 1)
        s = s + row_bcast<1>(t);
    if (n > 2)
        s = s + row_bcast<2>(t);
    if (n > 3)
        s = s + row_bcast<3>(t);
    return s;
}

// Test hook P.remote_delay (MC3HIP_TEST_REMOTE_DELAY_NS) for these loops, peer to peer: the
// rows of a refresh (all 64 lanes' worth) wait in a small LDS ring of the wave's own and go
// into the other ranks' replicas, in order, at the first publish point -- or poll -- after
// their time has come, so that the hook delays rows without holding the wave (handoff.h
// DelayedRows, for the wave's T chains at once).  Without the hook rows are stored at once.
constexpr int ROWS_RING = 8;                 // refreshes a wave may be ahead of its held-back rows
struct RowsLate {
    int handed, sent;
    unsigned stamp;                 // lane s: when the rows in slot s were produced
    size_t zoff, stride;
    double *ring;                   // LDS [ROWS_RING][64]
    __device__ __forceinline__ void begin(const DevProblem &P, long long R0, size_t lane_off,
                                          double *lds)
    {
        handed = sent = 0;
        stamp = 0;
        zoff = (size_t)(P.M0 + R0 * P.nchains) * ZROW + lane_off;
        stride = (size_t)P.nchains * ZROW;
        ring = lds;
    }
    __device__ __forceinline__ void drain(const DevProblem &P, const PeerTable<double> &peers,
                                          int lane, bool keeps, bool all)
    {
        const unsigned delay = (unsigned)P.remote_delay;
        while (sent < handed) {
            const unsigned t = (unsigned)__builtin_amdgcn_readlane((int)stamp, sent % ROWS_RING);
            if (realtime_ticks() - t < delay) {
                if (!all && handed - sent < ROWS_RING)
                    break;
                continue;            // (the end of the launch, or a full ring: wait for it)
            }
            const double v = ring[(sent % ROWS_RING) * 64 + lane];
            if (keeps)
                peers.store(zoff, v, P.rank);
            ++sent;
            zoff += stride;
        }
    }
    __device__ __forceinline__ void hand(const DevProblem &P, const PeerTable<double> &peers,
                                         int lane, bool keeps, double xk)
    {
        ring[(handed % ROWS_RING) * 64 + lane] = xk;
        const unsigned now = realtime_ticks();
        stamp = (lane == handed % ROWS_RING) ? now : stamp;
        ++handed;
        __builtin_amdgcn_wave_barrier();
        drain(P, peers, lane, keeps, false);
    }
};

// What the three samplers' rows loops share: the wave's chains and their states, the data
// sums of T proposals, the bookkeeping of an accepted step, the Z append.
// (FX: some coefficients are fixed or shared, or carry Gaussian priors -- lane 16 r + k holds
// FREE parameter k of its chain, coefficient j is free parameter kf[j] or the fixed value fx[j]
// (handoff.h coefficient_map), lane 16 r + g works out prior g of its chain)
// (P2P: the ensemble is sharded over GPUs -- rows are read from the local replica at system
// scope and go into every rank's replica, the chain wave's own stores: a refresh of T chains
// lasts longer than a remote store's acknowledgement, so no courier wave is needed; the
// log_post columns go out sixteen refreshes at a time)
template <int MT, int T, bool FX = false, bool P2P = false>
struct RowsWave {
    static_assert(T >= 2 && T <= 4, "2..4 chains per wave");
    static_assert(MT >= 2 && MT <= 4, "polynomials of 2..4 coefficients");
    static constexpr int ZS = ZROW;
    int lane, row, k, slot, kk, l, c, n, nacc, nf;
    int kf[MT];
    double fx[MT];
    PriorLanes<MT> priors;
    bool owner, exists, mine, keeps, first;
    double xk, cur, best, pmin_k, pmax_k;
    unsigned long long rowmask;
    DataTile<1, MT> tile;
    double *zblock, *lp_row;
    int32_t *zc_row;
    size_t blockstride;
    // P2P: log_post of the last (up to) 16 refreshes waits in lane 16 r + (refresh mod 16) of
    // its chain's row and is written 16 at a time: nobody reads it inside the launch
    double lpb;
    int ndone;
    PeerTable<double> peers_z;
    PeerTableLazy<double> peers_lp;
    RowsLate late;                  // (P2P, test hook: rows whose time has not come)

    // the first lane of the rows of chain q
    static constexpr int lane_of(int q) { return 16 * (T == 2 ? 2 * q : q); }

    __device__ __forceinline__ void init(const DevProblem &P, long long R0, double *lds_late = nullptr)
    {
        lane = threadIdx.x;
        row = lane >> 4;
        k = lane & 15;
        slot = (T == 2) ? row >> 1 : (row < T ? row : T - 1);      // the chain of this row
        owner = (T == 2) ? (row & 1) == 0 : row < T;               // (not a duplicate row)
        const int G = (int)gridDim.x, b = (int)blockIdx.x;
        tile.load(P);
        n = P.nchains;
        blockstride = (size_t)n * ZS;
        const double *pbase = (R0 == 0) ? P.snap0 : P.Z + (size_t)(P.M0 + (R0 - 1) * n) * ZS;
        zblock = P.Z + (size_t)(P.M0 + R0 * n) * ZS;
        lp_row = P.log_post + (P.M0 + R0 * n);
        zc_row = P.zchain + (P.M0 + R0 * n);
        // (the last chain of the wave may not exist: its rows store nothing)
        exists = b + slot * G < P.nlocal;
        l = exists ? b + slot * G : b;
        c = P.chain0 + l;
        nf = FX ? P.nfree : MT;
        if constexpr (FX) {
            coefficient_map<MT>(P, kf, fx);
            priors.init(P, kf, fx, k);
        }
        mine = k < nf;
        keeps = mine && owner && exists;              // lanes that store coefficients
        first = k == 0 && owner && exists;            // one lane per chain: the scalars
        kk = mine ? k : 0;
        xk = mine ? pbase[(size_t)c * ZS + k] : 0.0;
        cur = P.chisq[l];
        best = P.best_chisq[l];
        nacc = 0;
        pmin_k = mine ? P.fmin[k] : 0.0;
        pmax_k = mine ? P.fmax[k] : 0.0;
        rowmask = 0xffffull << (16 * row);
        lpb = 0.0;
        ndone = 0;
        if constexpr (P2P) {
            peers_z.load(P.peerZ, P.npeers);
            peers_lp.load(P.peerLP, P.npeers);
            if (P.remote_delay != 0) {
                for (int i = lane; i < ROWS_RING * 64; i += 64)
                    lds_late[i] = 0.0;
                late.begin(P, R0, (size_t)c * ZS + k, lds_late);
            }
        }
    }
    // (test hook: the held-back rows keep going out while this wave waits for somebody's --
    // ranks that wait for each other's held-back rows would never meet otherwise)
    __device__ __forceinline__ void while_waiting(const DevProblem &P)
    {
        if constexpr (P2P) {
            if (__builtin_expect(P.remote_delay != 0, 0))
                late.drain(P, peers_z, lane, keeps, false);
        }
    }
    // the chain of batch lane `q` (draws are made for the chains, not for the rows)
    __device__ __forceinline__ int chain_of(const DevProblem &P, int q) const
    {
        const int G = (int)gridDim.x, b = (int)blockIdx.x;
        return P.chain0 + ((b + q * G < P.nlocal) ? b + q * G : b);
    }
    // some lane of this lane's row says so
    __device__ __forceinline__ bool row_any(bool p) const { return (__ballot(p) & rowmask) != 0; }

    // boundaries, chain.py:235-243: true in the rows whose proposal has a coefficient out of
    // bounds (such a chain stays where it is: its data sum is computed and ignored)
    __device__ __forceinline__ bool out_of_bounds(const DevProblem &P, double npk) const
    {
        const bool outk = mine && (npk < pmin_k || npk > pmax_k);
        const bool row_out = row_any(outk);
        if (__builtin_expect(outk && owner && exists, 0))
            atomicAdd(&P.outbounds[k], 1ull);                            // chain.py:242-243
        return row_out;
    }
    __device__ __forceinline__ Model<MT> model_of(const DevProblem &P, double npk, int q) const
    {
        Model<MT> model;
        model.coef = nullptr;
        model.indep = P.indep;
        model.nmodel = P.nmodel;
        model.ndata = P.npad;
        // (v_readlane + moves; through the LDS crossbar instead -- ds_bpermute, 28 vector
        // instructions fewer per refresh -- measured 2 % SLOWER: the residuals wait for it)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            if constexpr (FX) {
                const double v = read_lane(npk, lane_of(q) + (kf[j] < 0 ? 0 : kf[j]));
                model.cf[j] = kf[j] < 0 ? fx[j] : v;
            } else {
                model.cf[j] = read_lane(npk, lane_of(q) + j);
            }
        }
        return model;
    }
    // the sum of lanes 0..nf-1 of every row, in lane order
    __device__ __forceinline__ double sum_free(double t) const
    {
        if constexpr (FX)
            return row_serial_sum_n(t, nf);
        else
            return row_serial_sum<MT>(t);
    }
    // PriorLanes::term for every row: lane 16 r + g works out prior g of the chain of row r
    // (the proposed value through the crossbar, the division), the squares added in the
    // priors' order
    __device__ __forceinline__ double prior_term(double npk) const
    {
        const double got = lane_gather(npk, (lane & 48) + (priors.src < 0 ? k : priors.src));
        const double v = priors.src >= 0 ? got : priors.fixedv;
        const double d = v - priors.centre;
        const double sd = (d > 0.0) ? d / priors.up : d / priors.lo;
        return row_serial_sum_n(sd * sd, priors.n);
    }
    // the data sums of the T proposals, chain.py:249: sixteen residuals per lane and chain,
    // then all the chains' trees at once; every lane of a row receives its chain's
    __device__ __forceinline__ double chisq_all(const DevProblem &P, double npk) const
    {
        double share[T];
#pragma unroll
        for (int q = 0; q < T; ++q)
            share[q] = gauss_lane_share<MT>(model_of(P, npk, q), tile);
        double total = rows_sum<T>(share);
        // (the "+inf model => chisq = inf" trap, resolved lazily as in block_gauss_model: only a
        // non-finite total has its model values looked at)
        const bool bad = (mc3_bits(total) & 0x7ff0000000000000ull) == 0x7ff0000000000000ull;
        const unsigned long long badm = __ballot(bad);
        if (__builtin_expect(badm != 0, 0)) {
#pragma unroll
            for (int q = 0; q < T; ++q) {
                if (!((badm >> lane_of(q)) & 1))
                    continue;
                const Model<MT> model = model_of(P, npk, q);
                int saw = 0;
                for (int i = lane; i < P.ndata; i += 64)
                    saw |= is_pos_inf(model.at(i));
                if (__any(saw))
                    total = (slot == q) ? mc3_from_bits(0x7ff0000000000000ull) : total;
            }
        }
        if constexpr (FX)                            // stats.py:211-216, stats.h:96-106
            total = total + prior_term(npk);
        return total;
    }
    // chain.py:261-289: the accepted rows move, the best fit so far, the Z append -- the T rows
    // in ONE store (the publish: sc1)
    __device__ __forceinline__ void commit(const DevProblem &P, bool acc, double npk,
                                           double nextchisq, long long R)
    {
        xk = acc ? npk : xk;
        // (the publish first -- everybody's next refresh waits for it -- the books after)
        // (issued on every path, the lanes chosen by EXEC: the loops that ask for the next
        // refresh's rows outside the compiler's wait counts retire them with vmcnt(1) BEHIND
        // this store -- handoff.h store_published_counted)
        store_published_counted<P2P>(zblock + (size_t)c * ZS + k, xk, keeps);
        const bool better = acc && nextchisq < best;
        if (__builtin_expect(__any(better), 0)) {
            if (better) {
                best = nextchisq;
                if (keeps)
                    P.bestx[(size_t)l * nf + k] = xk;
                if (first)
                    P.best_iter[l] = R;
            }
        }
        cur = acc ? nextchisq : cur;
        nacc += acc ? 1 : 0;
    }
    // P2P: the rows just published, into the other ranks' replicas (test hook P.remote_delay,
    // MC3HIP_TEST_REMOTE_DELAY_NS: through the ring, when their time has come)
    __device__ __forceinline__ void publish_remote(const DevProblem &P)
    {
        if constexpr (P2P) {
            if (__builtin_expect(P.remote_delay != 0, 0))
                late.hand(P, peers_z, lane, keeps, xk);
            else if (keeps)
                peers_z.store((size_t)(zblock - P.Z) + (size_t)c * ZS + k, xk, P.rank);
        }
    }
    // the column of log_post / zchain values waiting in the rows' lanes: the first `count`
    __device__ __forceinline__ void flush_columns(const DevProblem &P, int count) const
    {
        if (k < count && owner && exists) {
            const size_t off = (size_t)k * n + c;
            if constexpr (P2P)
                peers_lp.store((size_t)(lp_row - P.log_post) + off, lpb);
            else
                lp_row[off] = lpb;
            zc_row[off] = c;
        }
    }
    // log posterior and chain index of the new rows; on to the next refresh
    __device__ __forceinline__ void advance(const DevProblem &P)
    {
        if constexpr (!P2P) {
            // (one GPU: written as they come -- the column buffer below measured 1 % slower at
            // 8192 chains; peer to peer it saves a remote store per peer and refresh)
            if (first) {
                lp_row[c] = -0.5 * cur;
                zc_row[c] = c;
            }
            zblock += blockstride;
            lp_row += n;
            zc_row += n;
            return;
        }
        const int s = ndone & 15;
        lpb = (k == s) ? -0.5 * cur : lpb;
        if (__builtin_expect(s == 15, 0)) {
            flush_columns(P, 16);
            lp_row += (size_t)16 * n;
            zc_row += (size_t)16 * n;
        }
        zblock += blockstride;
        ndone += 1;
    }
    __device__ __forceinline__ void finish(const DevProblem &P)
    {
        if constexpr (P2P) {
            if (P.remote_delay != 0)
                late.drain(P, peers_z, lane, keeps, true);
        }
        if constexpr (P2P)
            flush_columns(P, ndone & 15);           // the unwritten tail (also after an early exit)
        if (keeps)
            P.X[(size_t)l * nf + k] = xk;
        if (first) {
            P.chisq[l] = cur;
            P.best_chisq[l] = best;
            P.numaccept[l] = P.numaccept[l] + nacc;
        }
    }
};

// ---------------------------------------------------------------------------------
// DEMC (lean_demc_mc_kernel's shape): the partner rows are rows of the refresh before, asked
// for at the top of a refresh and waited for; lane r T + q of a batch holds the draws of
// refresh bbase + r, chain q of the wave (chain.py:221-232, :259).
template <int MT, int T, bool FX = false, bool P2P = false>
__global__ __launch_bounds__(64) void lean_demc_rows_kernel(const DevProblem P, const int nref)
{
    constexpr int ZS = ZROW;
    constexpr int NB = 64 / T;                 // refreshes per batch of draws
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(
            &ctl_rw[CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
        return;
    const long long R0 = P.ctl[0];
    __shared__ double s_late[P2P ? ROWS_RING * 64 : 1];
    RowsWave<MT, T, FX, P2P> W;
    W.init(P, R0, s_late);
    const int lane = W.lane, n = W.n;
    const double *pbase = (R0 == 0) ? P.snap0 : P.Z + (size_t)(P.M0 + (R0 - 1) * n) * ZS;
    const double gamma = P.gamma, feps = P.fepsilon;
    const long long abort_at =
        (P.test_abort >= 0 && blockIdx.x == 0) ? (long long)P.test_abort : -1;
    bool alive = true;
    const int br = (lane / T < NB) ? lane / T : NB - 1;
    const int bc = W.chain_of(P, lane % T);
    double bu, blo, bhi;
    int bo1, bo2;
    long long bbase;
    auto fill = [&](long long base) {
        const Draw2 dd = draw2(P.seed, (uint64_t)(base + br), (uint32_t)bc, 0u);
        bu = u01(dd.q1);
        accept_window(bu, blo, bhi);
        int r1, r2;
        demc_partners(dd.q0, n, bc, r1, r2);
        bo1 = r1 * ZS;
        bo2 = r2 * ZS;
        bbase = base;
    };
    fill(R0);
    // this row's draws of a refresh, through the crossbar -- fetched a refresh AHEAD, in the
    // shadow of the wait for the partner rows, so that a refresh's requests leave the moment it
    // starts instead of a crossbar round trip later (the loop is bound by the hand-off)
    int o1, o2;
    double u, lo, hi;
    auto fetch = [&](long long R) {
        const int from = 4 * ((int)(R - bbase) * T + W.slot);    // whose draws this row needs
        o1 = __builtin_amdgcn_ds_bpermute(from, bo1) + W.kk;
        o2 = __builtin_amdgcn_ds_bpermute(from, bo2) + W.kk;
        u = gather_f64(bu, from);
        lo = gather_f64(blo, from);
        hi = gather_f64(bhi, from);
    };
    fetch(R0);
    const long long Rend = R0 + nref;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (long long R = R0; R < Rend; ++R) {
        if (__builtin_expect(R == abort_at, 0)) {                                         // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            break;                // (the host restores its checkpoint after an abort)
        }
        // (a pause in front of the polls, as in lean_demc_kernel -- the stores of the refresh
        // before clear the memory pipeline, and the partners' rows are likelier there at the first
        // look.  One GPU, 64 / 192 / 320 / 512 / 768 / 1 024 cycles: 4 096 chains 1 978 / 1 979 /
        // 1 974 / 1 990-2 002 / 2 001 / 1 909 M chain-steps/s, 8 192 chains 2 593 / 2 638 / 2 654 /
        // 2 679-2 688 / 2 666 / 2 603 -- 512 since round 6; peer to peer the stores are the
        // courier's or go to other devices: 64 as before)
        __builtin_amdgcn_s_sleep(P2P ? 1 : 8);
        // the partner rows of every chain of the wave: one request each
        const double *addr[2];
        addr[0] = pbase + o1;
        addr[1] = pbase + o2;
        double val[2];
        val[0] = load_published<P2P>(addr[0]);
        val[1] = load_published<P2P>(addr[1]);
        const double u_now = u, lo_now = lo, hi_now = hi;
        __builtin_amdgcn_sched_barrier(0);
        if (__builtin_expect(R - bbase == NB - 1, 0)) {
            // ... and behind them, once per batch, the draws of the next NB refreshes
            fill(R + 1);
            asm volatile("; pin" : "+v"(bu), "+v"(blo), "+v"(bhi), "+v"(bo1), "+v"(bo2));
        }
        fetch(R + 1);
        asm volatile("; draws of the next refresh" : "+v"(o1), "+v"(o2), "+v"(u), "+v"(lo), "+v"(hi));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        // what is still the sentinel is asked for again (the rows of the refresh before are the
        // last the other waves have published)
        if (__builtin_expect(__any(mc3_bits(val[0]) == ROW_PENDING ||
                                   mc3_bits(val[1]) == ROW_PENDING), 0)) {
            alive = wait_rows<2, P2P>(addr, val, ctl_rw, nullptr, [&]() { W.while_waiting(P); });
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("; rows of this refresh" : "+v"(val[0]), "+v"(val[1]));
            if (__builtin_expect(!alive, 0))
                break;
        }
        // proposal, chain.py:232
        const double jump_k = gamma * (val[0] - val[1]) + feps * 0.0;
        const double npk = W.xk + jump_k;
        const bool row_out = W.out_of_bounds(P, npk);
        const double nextchisq = W.chisq_all(P, npk);
        // Metropolis, chain.py:257-274, every chain in its own rows
        const bool acc = row_out ? false
                                 : metropolis_accept(0.5 * (W.cur - nextchisq), u_now, lo_now,
                                                     hi_now);
        W.commit(P, acc, npk, nextchisq, R);
        W.publish_remote(P);
        pbase = W.zblock;
        W.advance(P);
    }
    W.finish(P);
}

// ---------------------------------------------------------------------------------
// DEMC in the DECLARED MULTI-GPU CADENCE (mc3hip_problem.zlag = D >= 1, lean_demc_lag_kernel's
// shape): during refresh R the chains see the others through the rows of refresh R - 1 - D
// (the initial snapshot while that is negative), published at least a refresh ago -- so the
// partner rows of the NEXT refresh are asked for at the top of this one by loads outside the
// compiler's wait counts (handoff.h load_published_async), retired behind this refresh's
// publish, and no refresh waits for a round trip.  Same draws, same arithmetic, same results
// as run_chain's with the same zlag.
template <int MT, int T, bool FX = false, bool P2P = false>
__global__ __launch_bounds__(64) void lean_demc_rows_lag_kernel(const DevProblem P, const int nref)
{
    constexpr int ZS = ZROW;
    constexpr int NB = 64 / T;                 // refreshes per batch of draws
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(
            &ctl_rw[CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
        return;
    const long long R0 = P.ctl[0];
    __shared__ double s_late[P2P ? ROWS_RING * 64 : 1];
    RowsWave<MT, T, FX, P2P> W;
    W.init(P, R0, s_late);
    const int lane = W.lane, n = W.n;
    const long long D = P.zlag;
    // the rows refresh R looks at: of refresh R - 1 - D
    auto seen_by = [&](long long R) -> const double * {
        const long long Rs = R - 1 - D;
        return (Rs < 0) ? P.snap0 : P.Z + (size_t)(P.M0 + Rs * n) * ZS;
    };
    const double gamma = P.gamma, feps = P.fepsilon;
    const long long abort_at =
        (P.test_abort >= 0 && blockIdx.x == 0) ? (long long)P.test_abort : -1;
    const int br = (lane / T < NB) ? lane / T : NB - 1;
    const int bc = W.chain_of(P, lane % T);
    double bu, blo, bhi;
    int bo1, bo2;
    long long bbase;
    auto fill = [&](long long base) {
        const Draw2 dd = draw2(P.seed, (uint64_t)(base + br), (uint32_t)bc, 0u);
        bu = u01(dd.q1);
        accept_window(bu, blo, bhi);
        int r1, r2;
        demc_partners(dd.q0, n, bc, r1, r2);
        bo1 = r1 * ZS;
        bo2 = r2 * ZS;
        bbase = base;
    };
    struct Step {
        double u, lo, hi;
        const double *addr[2];
        double val[2];
    };
    // (asked for at ONE place per refresh, waited for at one: handoff.h load_published_async)
    auto ask = [&](Step &S) {
        S.val[0] = load_published_async<P2P>(S.addr[0]);
        S.val[1] = load_published_async<P2P>(S.addr[1]);
    };
    auto prepare = [&](long long R, Step &S) {
        const int from = 4 * ((int)(R - bbase) * T + W.slot);    // whose draws this row needs
        const double *sb = seen_by(R);
        S.addr[0] = sb + (__builtin_amdgcn_ds_bpermute(from, bo1) + W.kk);
        S.addr[1] = sb + (__builtin_amdgcn_ds_bpermute(from, bo2) + W.kk);
        S.u = gather_f64(bu, from);
        S.lo = gather_f64(blo, from);
        S.hi = gather_f64(bhi, from);
    };
    fill(R0);
    Step S;
    prepare(R0, S);
    ask(S);
    {
        double none = 0.0;
        published_arrived<0>(S.val[0], S.val[1], none);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const long long Rend = R0 + nref;
    bool alive = true;
    for (long long R = R0; R < Rend; ++R) {
        if (__builtin_expect(R == abort_at, 0)) {                                         // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        // the rows of this refresh were asked for at the top of the one before and retired at
        // its bottom; the next refresh's are asked for now
        Step C = S;
        Step Nx;
        if (__builtin_expect(R + 1 - bbase == NB, 0))
            fill(R + 1);
        prepare(R + 1, Nx);
        ask(Nx);
        // (D = 1: a row of the refresh before may still be the sentinel)
        if (__builtin_expect(__any(mc3_bits(C.val[0]) == ROW_PENDING ||
                                   mc3_bits(C.val[1]) == ROW_PENDING), 0)) {
            const double *const a2[2] = {C.addr[0], C.addr[1]};
            double v2[2] = {C.val[0], C.val[1]};
            alive = wait_rows<2, P2P>(a2, v2, ctl_rw, nullptr, [&]() { W.while_waiting(P); });
            // (nothing this path asked for is pending where it joins the other)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("; rows of this refresh" : "+v"(v2[0]), "+v"(v2[1]));
            C.val[0] = v2[0];
            C.val[1] = v2[1];
            if (__builtin_expect(!alive, 0))
                break;
        }
        const double jump_k = gamma * (C.val[0] - C.val[1]) + feps * 0.0;     // chain.py:232
        const double npk = W.xk + jump_k;
        const bool row_out = W.out_of_bounds(P, npk);
        const double nextchisq = W.chisq_all(P, npk);
        const bool acc = row_out ? false
                                 : metropolis_accept(0.5 * (W.cur - nextchisq), C.u, C.lo, C.hi);
        W.commit(P, acc, npk, nextchisq, R);
        // the next refresh's rows: everything but the youngest operation of this wave -- the
        // publish above -- has completed
        {
            double none = 0.0;
            published_arrived<1>(Nx.val[0], Nx.val[1], none);
        }
        W.publish_remote(P);                       // (behind the wait: nothing waits for these)
        S = Nx;
        W.advance(P);
    }
    W.finish(P);
}

// ---------------------------------------------------------------------------------
// The random walk (lean_mrw_kernel's arithmetic): nothing is read.  Lane (3 i + s) T + q of a
// batch holds Philox slot s' (0, 3, 4 for s = 0, 1, 2) of refresh bbase + i, chain q: the
// Metropolis draw with its window from s = 0, the pairs of normal deviates of coefficients
// 0-1 and 2-3 from s = 1, 2 (lean_mrw_kernel's fields).
template <int MT, int T, bool FX = false, bool P2P = false>
__global__ __launch_bounds__(64) void lean_mrw_rows_kernel(const DevProblem P, const int nref)
{
    constexpr int NB = 64 / (3 * T);           // refreshes per batch of draws
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(
            &ctl_rw[CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
        return;
    const long long R0 = P.ctl[0];
    __shared__ double s_late[P2P ? ROWS_RING * 64 : 1];
    RowsWave<MT, T, FX, P2P> W;
    W.init(P, R0, s_late);
    const int lane = W.lane;
    const double pstep_k = W.mine ? P.fstep[W.k] : 0.0;
    const long long abort_at =
        (P.test_abort >= 0 && blockIdx.x == 0) ? (long long)P.test_abort : -1;
    const int bi = (lane / (3 * T) < NB) ? lane / (3 * T) : NB - 1;
    const int bs = (lane / T) % 3;
    const int bc = W.chain_of(P, lane % T);
    double bu, blo, bhi, bcs, bsn;
    long long bbase = R0;
    auto fill = [&](long long base) {
        const Draw2 d = draw2(P.seed, (uint64_t)(base + bi), (uint32_t)bc,
                              bs == 0 ? 0u : (uint32_t)(2 + bs));
        bu = u01(d.q1);                                                  // chain.py:259
        accept_window(bu, blo, bhi);
        const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(d.q0)));
        double sn, cs;
        mc3_det_sincos2pi(u01(d.q1), &sn, &cs);
        bcs = rad * cs;
        bsn = rad * sn;
        bbase = base;
    };
    fill(R0);
    // this lane's fields of a refresh: lane 3 T i + q (s = 0), 3 T i + (1 + k / 2) T + q
    const int at0 = 4 * W.slot;
    const int at1 = W.mine ? 4 * ((1 + (W.k >> 1)) * T + W.slot) : at0;
    const long long Rend = R0 + nref;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (long long R = R0; R < Rend; ++R) {
        if (__builtin_expect(R == abort_at, 0)) {                                         // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        if (__builtin_expect(R - bbase == NB, 0))
            fill(R);
        const int base = 4 * 3 * T * (int)(R - bbase);
        const double u = gather_f64(bu, base + at0), lo = gather_f64(blo, base + at0),
                     hi = gather_f64(bhi, base + at0);
        const double gc = gather_f64(bcs, base + at1), gs = gather_f64(bsn, base + at1);
        const double jump_k = W.mine ? ((W.k & 1) ? gs : gc) * pstep_k : 0.0;
        const double npk = W.xk + jump_k;
        const bool row_out = W.out_of_bounds(P, npk);
        const double nextchisq = W.chisq_all(P, npk);
        const bool acc = row_out ? false
                                 : metropolis_accept(0.5 * (W.cur - nextchisq), u, lo, hi);
        W.commit(P, acc, npk, nextchisq, R);
        W.publish_remote(P);
        W.advance(P);
    }
    W.finish(P);
}

// ---------------------------------------------------------------------------------
// Snooker (lean_snooker_kernel's arithmetic, chain.py:195-220, :251-255): three rows of the
// history per chain and refresh -- nearly always published long ago -- asked for one REFRESH
// ahead by loads outside the compiler's wait counts (handoff.h load_published_async) and
// retired behind the publish of the refresh between.  Lane (3 i + s) T + q of a batch holds
// Philox slot s of refresh bbase + i, chain q (lean_snooker_kernel's fields: s = 0 the two
// rows of the difference, s = 1 the row z and whether this is a snooker jump, s = 2 its scale
// and the Metropolis draw with its window).  A refresh in which no chain of the wave makes a
// snooker jump (two in three at four chains) never enters that path; in one that does, the
// rows that do not are masked out of it.
template <int MT, int T, bool FX = false, bool P2P = false>
__global__ __launch_bounds__(64) void lean_snooker_rows_kernel(const DevProblem P, const int nref)
{
    constexpr int ZS = ZROW;
    constexpr int NB = 64 / (3 * T);
    long long *ctl_rw = const_cast<long long *>(P.ctl);
    if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(
            &ctl_rw[CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
        return;
    const long long R0 = P.ctl[0];
    __shared__ double s_late[P2P ? ROWS_RING * 64 : 1];
    RowsWave<MT, T, FX, P2P> W;
    W.init(P, R0, s_late);
    const int lane = W.lane, n = W.n;
    const double gamma = P.gamma, feps = P.fepsilon;
    const long long abort_at =
        (P.test_abort >= 0 && blockIdx.x == 0) ? (long long)P.test_abort : -1;
    const int bi = (lane / (3 * T) < NB) ? lane / (3 * T) : NB - 1;
    const int bs = (lane / T) % 3;
    const int bc = W.chain_of(P, lane % T);
    unsigned long long A, B;
    double blo, bhi;
    long long bbase = R0;
    const long long zlag = P.zlag;
    auto fill = [&](long long base) {
        const long long Rl = base + bi;
        // (the history a refresh draws from: the rows before it -- zlag refreshes before it in
        // the declared multi-GPU cadence, mc3hip_problem.zlag)
        const long long Rh = Rl - zlag;
        const unsigned long long zsz = (unsigned long long)(P.M0 + (Rh > 0 ? Rh : 0) * n);
        const Draw2 d = draw2(P.seed, (uint64_t)Rl, (uint32_t)bc, (uint32_t)bs);
        if (bs == 0) {
            const unsigned long long iR1 = bounded(d.q0, zsz);
            unsigned long long iR2 = 1 + bounded(d.q1, zsz - 1);
            if (iR2 == iR1)
                iR2 = 0;
            A = iR1 * ZS;
            B = iR2 * ZS;
        } else if (bs == 1) {
            A = bounded(d.q1, zsz) * ZS;
            B = (u01(d.q0) < 0.1) ? 1ull : 0ull;
        } else {
            A = mc3_bits(1.2 + (2.2 - 1.2) * u01(d.q0));
            B = mc3_bits(u01(d.q1));
        }
        accept_window(u01(d.q1), blo, bhi);
        bbase = base;
    };
    struct Step {
        int sj;
        double gs, u, lo, hi;
        const double *addr[3];
        double val[3];
    };
    // (the rows are asked for by ask(), at ONE place per refresh, and waited for at one:
    // registers that loads the compiler does not see are on their way to must not meet other
    // values at a join of two paths)
    auto ask = [&](Step &S) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            S.val[j] = load_published_async<P2P>(S.addr[j]);
    };
    auto prepare = [&](long long R, Step &S) {
        const int base = 4 * (3 * T * (int)(R - bbase) + W.slot);
        S.addr[0] = P.Z + (gather_u64(A, base) + (unsigned)W.kk);
        S.addr[1] = P.Z + (gather_u64(B, base) + (unsigned)W.kk);
        S.addr[2] = P.Z + (gather_u64(A, base + 4 * T) + (unsigned)W.kk);
        S.sj = __builtin_amdgcn_ds_bpermute(base + 4 * T, (int)(unsigned)B);
        S.gs = mc3_from_bits(gather_u64(A, base + 8 * T));
        S.u = mc3_from_bits(gather_u64(B, base + 8 * T));
        S.lo = gather_f64(blo, base + 8 * T);
        S.hi = gather_f64(bhi, base + 8 * T);
    };
    fill(R0);
    Step S;
    prepare(R0, S);
    ask(S);
    published_arrived<0>(S.val[0], S.val[1], S.val[2]);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const long long Rend = R0 + nref;
    bool alive = true;
    for (long long R = R0; R < Rend; ++R) {
        if (__builtin_expect(R == abort_at, 0)) {                                         // test hook
            __hip_atomic_store(&ctl_rw[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        // the rows of this refresh were asked for at the top of the one before and retired at
        // its bottom; the next refresh's are asked for now
        Step C = S;
        Step Nx;
        if (__builtin_expect(R + 1 - bbase == NB, 0))
            fill(R + 1);
        prepare(R + 1, Nx);
        ask(Nx);
        // (one of the newest rows may still be the sentinel)
        if (__builtin_expect(__any(mc3_bits(C.val[0]) == ROW_PENDING ||
                                   mc3_bits(C.val[1]) == ROW_PENDING ||
                                   mc3_bits(C.val[2]) == ROW_PENDING), 0)) {
            const double *const a3[3] = {C.addr[0], C.addr[1], C.addr[2]};
            double v3[3] = {C.val[0], C.val[1], C.val[2]};
            alive = wait_rows<3, P2P>(a3, v3, ctl_rw, nullptr, [&]() { W.while_waiting(P); });
            // (nothing this path asked for is pending where it joins the other)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("; rows of this refresh" : "+v"(v3[0]), "+v"(v3[1]), "+v"(v3[2]));
            C.val[0] = v3[0];
            C.val[1] = v3[1];
            C.val[2] = v3[2];
            if (__builtin_expect(!alive, 0))
                break;
        }
        const double z1 = C.val[0], z2 = C.val[1], zk = C.val[2];
        const bool sjump = C.sj != 0;                // (the whole row says the same)
        const bool any_sjump = __any(sjump);
        double jump_k = gamma * (z1 - z2) + feps * 0.0;                  // chain.py:216-217
        if (__builtin_expect(any_sjump, 0)) {                            // chain.py:207-215
            const bool same = !W.row_any(W.mine && !(zk == W.xk));
            const double dz = W.xk - zk;
            const double zp1 = W.sum_free(z1 * dz);
            const double zp2 = W.sum_free(z2 * dz);
            const double dd = W.sum_free(dz * dz);
            const double sj_k = same ? C.gs * (z2 - z1) : ((C.gs * (zp1 - zp2)) * dz) / dd;
            jump_k = sjump ? sj_k : jump_k;
        }
        const double npk = W.xk + jump_k;
        const bool row_out = W.out_of_bounds(P, npk);
        const double nextchisq = W.chisq_all(P, npk);
        bool acc = (row_out || sjump)
                       ? false
                       : metropolis_accept(0.5 * (W.cur - nextchisq), C.u, C.lo, C.hi);
        if (__builtin_expect(any_sjump, 0)) {                            // chain.py:251-255
            const double dc = W.xk - zk, dn = npk - zk;
            const double cnorm = W.sum_free(W.mine ? dc * dc : 0.0);
            const double nnorm = W.sum_free(W.mine ? dn * dn : 0.0);
            const double mrfactor = mc3_det_pow_half(nnorm / cnorm, W.nf - 1);
            const bool acc_s = mc3_det_exp(0.5 * (W.cur - nextchisq)) * mrfactor > C.u;
            acc = (sjump && !row_out) ? acc_s : acc;
        }
        W.commit(P, acc, npk, nextchisq, R);
        // the next refresh's rows: everything but the youngest operation of this wave -- the
        // publish above -- has completed
        published_arrived<1>(Nx.val[0], Nx.val[1], Nx.val[2]);
        W.publish_remote(P);
        S = Nx;
        W.advance(P);
    }
    W.finish(P);
}

}  // namespace mc3hip
