// handoff.h -- in-launch hand-off of Z rows between workgroups, the draws of an iteration.
#pragma once
#include "core.h"

namespace mc3hip {

// A Z row read through the in-launch hand-off (persistent kernel): rows are
// published with agent-scope 8-byte stores and consumed with agent-scope loads
// (sc1: they bypass the CU's L1, which another CU's stores never refresh); a row
// that is not there yet still holds the all-ones NaN pattern the Z buffer is
// filled with at creation, so every double is its own "ready" flag -- no separate
// flag, fence or ordering is needed (8-byte stores are single-copy atomic).
// A chain state can never legitimately be NaN: a NaN proposal has a NaN or
// out-of-bounds chisq and is rejected (chain.py:257-260).
// (ROW_PENDING, CTL_ABORT: core.h)

template <bool SYS = false>
__device__ __forceinline__ double load_published(const double *p)
{
    // SYS: the row may have been stored by another GPU (system scope, fine-grained
    // memory); otherwise by another CU of this GPU (agent scope)
    if constexpr (SYS)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A launch queued behind one that gave up must not run -- its inputs are missing -- and the
// decision is ONE per workgroup: thread 0 looks, everybody goes by what it saw.  (Each wave
// looking for itself can disagree with its neighbours when the word flips between their
// loads: the waves that stay then pass the barriers alone and wait for partners that left.)
// `word`: an int of LDS nobody writes before the barrier in here.
__device__ __forceinline__ bool launch_given_up(const long long *ctl, int *word)
{
    if (threadIdx.x == 0)
        *word = (__hip_atomic_load(&ctl[CTL_ABORT], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1 : 0;
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(*word) != 0;
}

// The same load issued from inline assembly: the compiler's wait-count bookkeeping does not
// know it, so no wait for it is inserted anywhere -- in particular not the `vmcnt(0)` the
// compiler puts at a loop head when loads and stores are both in flight, which would wait for
// the acknowledgement of the row just published.  The wave's vector-memory operations are
// counted in issue order, loads and stores alike (MI355X_MICROARCH.md, "s_waitcnt vmcnt(N)
// waits until all but the N youngest are done"), so the caller retires these loads with
// published_arrived<N>: N = operations it is SURE to have issued after them (at least).  The
// loaded-into variables must not be touched between the two (a copy would read a register the
// load has not written yet) -- and the compiler must have no reason to copy them either: ONE
// place that loads into a set of variables and ONE that waits for it, in straight-line code or
// with the loop's back edge behind the wait.  Two load sites on two paths that join, or two
// waits on two paths, make the variables meet other values at the join, where the compiler
// puts one of them into the other's register with a copy -- on an iteration short enough (a
// proposal out of bounds) the copy runs before the load has returned.  Found by a test
// (test_lean_snooker_and_mrw_loops_with_thinning[tight]); tools/waits.py shows the sites.
template <bool SYS = false>
__device__ __forceinline__ double load_published_async(const double *p)
{
    double v;
    // (SYS: system scope, as load_published<true> -- the row may have come over xGMI)
    if constexpr (SYS)
        asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=&v"(v) : "v"(p) : "memory");
    else
        asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void published_arrived(double &a, double &b, double &c)
{
    static_assert(N >= 0 && N <= 3, "");
    if constexpr (N == 0)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c)::"memory");
    else if constexpr (N == 1)
        asm volatile("s_waitcnt vmcnt(1)" : "+v"(a), "+v"(b), "+v"(c)::"memory");
    else if constexpr (N == 2)
        asm volatile("s_waitcnt vmcnt(2)" : "+v"(a), "+v"(b), "+v"(c)::"memory");
    else
        asm volatile("s_waitcnt vmcnt(3)" : "+v"(a), "+v"(b), "+v"(c)::"memory");
}

// The store such a hand-counted wait relies on ("everything but the youngest operation -- the
// store above at least -- has completed"): ISSUED ON EVERY PATH.  Written as `if (mine) store`
// the compiler masks the block and skips it with s_cbranch_execz -- never taken here (some lane
// always keeps a coefficient), but a path of the control-flow graph on which NOTHING follows
// the hidden loads, so that vmcnt(1) would retire none of them: mc3_amd/isa_audit.py counts
// operations per path (round 5) and rightly refuses it.  Here the lanes are chosen by EXEC
// set in the same inline assembly; there is no branch to take.
template <bool SYS>
__device__ __forceinline__ void store_published_counted(double *p, double v, bool mine)
{
    const uint64_t mask = __builtin_amdgcn_ballot_w64(mine);      // (a subset of EXEC)
    uint64_t saved;
    if constexpr (SYS)
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\t"
                     "global_store_dwordx2 %2, %3, off sc0 sc1\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved)
                     : "s"(mask), "v"(p), "v"(v)
                     : "memory");
    else
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\t"
                     "global_store_dwordx2 %2, %3, off sc1\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved)
                     : "s"(mask), "v"(p), "v"(v)
                     : "memory");
}

// The ranks' replicas (peer-to-peer mode).  The pointers are read ONCE, before the loop: a load
// of one inside it is waited for together with everything older in the wave's queue -- the
// store into the previous peer's replica, still on its way over xGMI -- so a row's stores
// would go out one acknowledged round trip after the other.  Up to PEER_REGS ranks (one
// node) are kept in scalar registers; ranks beyond that are served the slow way.
constexpr int PEER_REGS = 8;
template <class T>
struct PeerTable {
    T *p[PEER_REGS];
    T *const *table;
    int n;
    __device__ __forceinline__ void load(T *const *tab, int npeers)
    {
        table = tab;
        n = npeers;
#pragma unroll
        for (int r = 0; r < PEER_REGS; ++r) {
            const unsigned long long q =
                (unsigned long long)(r < npeers ? tab[r] : (T *)nullptr);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)q);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32));
            p[r] = own_global((T *)(((unsigned long long)hi << 32) | lo));
        }
    }
    // v to element `off` of every replica (but `skip`'s), at system scope
    __device__ __forceinline__ void store(size_t off, T v, int skip = -1) const
    {
#pragma unroll
        for (int r = 0; r < PEER_REGS; ++r)
            if (r < n && r != skip)
                __hip_atomic_store(p[r] + off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int r = PEER_REGS; r < n; ++r)
            if (r != skip)
                __hip_atomic_store(&table[r][off], v, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
    }
};

// The same for stores that are rare (the log_post columns, 64 refreshes at a time): the table
// stays in memory and is read when it is needed, so that no scalar registers are tied up for
// the whole loop (sixteen of them: with a second register-resident table the peer-to-peer
// loops spill scalars into vector lanes -- 246 registers, two waves per SIMD with nothing to
// spare, which several ranks sharing ONE device to its last wave slot have been seen not to
// survive).  The wait for the pointers lands behind stores that nobody polls.
template <class T>
struct PeerTableLazy {
    T *const *table;
    int n;
    __device__ __forceinline__ void load(T *const *tab, int npeers)
    {
        table = tab;
        n = npeers;
    }
    __device__ __forceinline__ void store(size_t off, T v) const
    {
        for (int r = 0; r < n; ++r)
            __hip_atomic_store(&table[r][off], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
};

// ---- the courier wave (peer-to-peer mode) ---------------------------------------------------
// A wave's loads and stores retire through ONE counter in issue order (MI355X_MICROARCH.md,
// s_waitcnt): a chain wave that stored its row into seven remote replicas would have the data
// of its next loads held back until every one of those stores is acknowledged over xGMI -- a
// round trip, where the hand-off needs one way.  So the chain wave stores into the LOCAL
// replica only and leaves the row in an LDS ring; a second wave of the workgroup, the courier,
// picks it up (it spins on LDS: no vector-memory operation, nothing the chain wave could wait
// behind), stores it into the remote replicas, its own counter absorbing the acknowledgements.
// An empty slot holds the sentinel no row can hold (as an unwritten row of Z does), so one LDS
// read tells the courier that the row is there and what it is.  s_cour: [1] rows sent (every
// eighth), [2] the chain wave has left, [3] go / no go.
constexpr int COURIER_RING = 64;             // rows the chain wave may be ahead of its courier

// both waves, first thing: empty ring, ONE decision for both (a courier that left while its
// chain wave stays -- an abort raised between their two looks at the word -- would never drain
// the ring).  false: the launch had been given up before it started
__device__ __forceinline__ bool courier_begin(double *s_ring, int *s_cour, long long *ctl_rw)
{
    if (threadIdx.x == 0) {
        s_cour[1] = 0;
        s_cour[2] = 0;
        s_cour[3] = __hip_atomic_load(&ctl_rw[CTL_ABORT], __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT) == 0;
    }
    for (int i = threadIdx.x; i < COURIER_RING * ZROW; i += 128)
        s_ring[i] = mc3_from_bits(ROW_PENDING);
    __syncthreads();
    return s_cour[3] != 0;
}

__device__ __forceinline__ unsigned realtime_ticks()       // 100 MHz, wraps: differences only
{
    return (unsigned)__builtin_amdgcn_s_memrealtime();
}

// the courier's loop: rows of chain c, refresh R0c onwards, from the ring to every replica but
// the local one (P2P; otherwise -- one GPU, an experiment -- to the only one).
// Test hook P.remote_delay (MC3HIP_TEST_REMOTE_DELAY_NS): a row is held back that long after it
// showed up in the ring -- several ranks on one GPU then see each other as if an xGMI hop
// apart.  Rows are still picked up at once and sent in order, each when ITS time has come (two
// cursors), so the hook delays rows without throttling them.
template <bool P2P>
__device__ __forceinline__ void courier_run(const DevProblem &P, double *s_ring, int *s_cour,
                                            const int lane, const int nf, const int c)
{
    constexpr int ZS = ZROW;
    const bool mine = lane < nf;
    const int kk = mine ? lane : 0;
    const long long R0c = P.ctl[0];
    size_t zoff = (size_t)(P.M0 + R0c * P.nchains) * ZS + (size_t)c * ZS + lane;
    const size_t stride = (size_t)P.nchains * ZS;
    PeerTable<double> peers;
    if constexpr (P2P)
        peers.load(P.peerZ, P.npeers);
    double *const zlocal = own_global(P.Z);
    (void)zlocal;
    const unsigned delay = (unsigned)P.remote_delay;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int sent = 0;
    bool last = false;
    auto send = [&](double *slot, double v) {
        if constexpr (P2P) {
            if (mine)
                peers.store(zoff, v, P.rank);
        } else {
            if (mine)
                __hip_atomic_store(zlocal + zoff, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (mine)
            __hip_atomic_store(slot + lane, mc3_from_bits(ROW_PENDING), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
        ++sent;
        zoff += stride;
        if ((sent & 7) == 0)
            __hip_atomic_store(&s_cour[1], sent, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    if (__builtin_expect(delay == 0, 1)) {
        for (;;) {
            double *const slot = s_ring + (sent % COURIER_RING) * ZS;
            const double v = __hip_atomic_load(slot + kk, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__any(mine && mc3_bits(v) == ROW_PENDING)) {
                // nothing yet.  Once the chain wave has left, one more look: its last row went
                // into the ring before the word that says so
                if (last)
                    break;
                last = __hip_atomic_load(&s_cour[2], __ATOMIC_ACQUIRE,
                                         __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
                continue;
            }
            last = false;
            send(slot, v);
        }
        return;
    }
    // the test hook: `seen` rows have shown up (lane s of `stamp`: when the row in slot s did)
    int seen = 0;
    unsigned stamp = 0;
    for (;;) {
        bool progress = false;
        if (seen - sent < COURIER_RING) {
            double *const slot = s_ring + (seen % COURIER_RING) * ZS;
            const double v = __hip_atomic_load(slot + kk, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!__any(mine && mc3_bits(v) == ROW_PENDING)) {
                const unsigned now = realtime_ticks();
                stamp = (lane == seen % COURIER_RING) ? now : stamp;
                ++seen;
                progress = true;
                last = false;
            }
        }
        if (sent < seen) {
            const unsigned t = (unsigned)__builtin_amdgcn_readlane((int)stamp, sent % COURIER_RING);
            if (realtime_ticks() - t >= delay) {
                double *const slot = s_ring + (sent % COURIER_RING) * ZS;
                const double v = __hip_atomic_load(slot + kk, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                send(slot, v);
                progress = true;
            }
        } else if (!progress) {
            if (last)
                break;
            last = __hip_atomic_load(&s_cour[2], __ATOMIC_ACQUIRE,
                                     __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
        }
    }
}

// the chain wave hands its new row to the courier (after its own store into the local replica);
// never a whole ring ahead: looked at once in 16 rows.  false: the launch is being given up
__device__ __forceinline__ bool courier_hand(double *s_ring, int *s_cour, int &handed,
                                             const int lane, const bool mine, const double xk,
                                             long long *ctl_rw)
{
    if (mine)
        __hip_atomic_store(&s_ring[(handed % COURIER_RING) * ZROW + lane], xk, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    ++handed;
    if (__builtin_expect((handed & 15) == 0, 0)) {
        while (handed - __hip_atomic_load(&s_cour[1], __ATOMIC_ACQUIRE,
                                          __HIP_MEMORY_SCOPE_WORKGROUP) > COURIER_RING - 16) {
            if (__hip_atomic_load(&ctl_rw[CTL_ABORT], __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT) != 0)
                return false;
            __builtin_amdgcn_s_sleep(2);
        }
    }
    return true;
}

// the chain wave leaves (normally, giving up or aborted): the courier drains the ring and goes
__device__ __forceinline__ void courier_end(int *s_cour)
{
    __hip_atomic_store(&s_cour[2], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The same test hook for a chain wave that stores into the remote replicas ITSELF (no courier:
// more than two ranks on one device, MC3HIP_NO_COURIER): the rows wait in an LDS ring of the
// wave's own and go out, in order, at the first publish point after their time has come
// (flush: at the end, after waiting for it).  Without the hook (P.remote_delay = 0) rows are
// stored at once and this is not used.
struct DelayedRows {
    int handed, sent;
    unsigned stamp;                 // lane s: when the row in slot s was produced
    size_t zoff, stride;
    __device__ __forceinline__ void begin(const DevProblem &P, long long R0, int c, int lane)
    {
        handed = sent = 0;
        stamp = 0;
        zoff = (size_t)(P.M0 + R0 * P.nchains) * ZROW + (size_t)c * ZROW + lane;
        stride = (size_t)P.nchains * ZROW;
    }
    __device__ __forceinline__ void drain(const DevProblem &P, double *s_ring,
                                          const PeerTable<double> &peers, int lane, bool mine,
                                          bool all)
    {
        const unsigned delay = (unsigned)P.remote_delay;
        while (sent < handed) {
            const unsigned t = (unsigned)__builtin_amdgcn_readlane((int)stamp, sent % COURIER_RING);
            if (realtime_ticks() - t < delay) {
                if (!all && handed - sent < COURIER_RING)
                    break;
                continue;            // (the end of the launch, or a full ring: wait for it)
            }
            const double v = s_ring[(sent % COURIER_RING) * ZROW + (mine ? lane : 0)];
            if (mine)
                peers.store(zoff, v, P.rank);
            ++sent;
            zoff += stride;
        }
    }
    __device__ __forceinline__ void hand(const DevProblem &P, double *s_ring,
                                         const PeerTable<double> &peers, int lane, bool mine,
                                         double xk)
    {
        if (mine)
            s_ring[(handed % COURIER_RING) * ZROW + lane] = xk;
        const unsigned now = realtime_ticks();
        stamp = (lane == handed % COURIER_RING) ? now : stamp;
        ++handed;
        __builtin_amdgcn_wave_barrier();
        drain(P, s_ring, peers, lane, mine, false);
    }
};

// ... and where the chain wave is held itself (the general chain, the owner of the resident
// tiled plans): the row goes into the local replica at once and into the others after the
// delay, the wave waiting it out -- a pessimistic model (a real store does not hold the wave,
// but its acknowledgement holds the wave's next loads back as long)
__device__ __forceinline__ void store_row_everywhere(const DevProblem &P,
                                                     const PeerTable<double> &peers, size_t off,
                                                     double v, bool mine)
{
    if (__builtin_expect(P.remote_delay != 0, 0)) {
        if (mine)
            __hip_atomic_store(P.Z + off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned t0 = realtime_ticks();
        while (realtime_ticks() - t0 < (unsigned)P.remote_delay)
            __builtin_amdgcn_s_sleep(1);
        if (mine)
            peers.store(off, v, P.rank);
    } else if (mine) {
        peers.store(off, v);
    }
}

// Wave-level wait until every lane's values have been published (all 64 lanes
// poll; lanes beyond nfree repeat lane 0's addresses).  The first read is issued
// by the caller so that independent scalar work can overlap its latency; this
// only spins when something is still pending.  Bounded: gives up (and tells
// everybody) after a few seconds so a lost workgroup can never hang the device;
// the host then restores its checkpoint and redoes the run (mc3hip.hip settle()).
struct NoIdle {
    __device__ __forceinline__ void operator()() const {}
};
// (`idle`: called once per round of polls -- the test hook's queue of held-back rows must keep
// draining while its wave waits, or ranks that wait for each other's held-back rows never meet)
template <int NVAL, bool SYS = false, class Idle = NoIdle>
__device__ __forceinline__ bool wait_rows(const double *const (&addr)[NVAL], double (&val)[NVAL],
                                          long long *ctl, unsigned long long *nspin = nullptr,
                                          Idle idle = Idle())
{
    for (unsigned spins = 0;; ++spins) {
        idle();
        if (nspin && spins)
            *nspin += 1;                      // diagnostic build: polls beyond the first
        bool pend[NVAL];
        bool any_pending = false;
#pragma unroll
        for (int q = 0; q < NVAL; ++q) {
            pend[q] = __any(mc3_bits(val[q]) == ROW_PENDING);
            any_pending |= pend[q];
        }
        if (!any_pending)
            return true;
        // (no pause, and only the rows still missing are asked for again: every extra request
        // on the fabric and every cycle of back-off measurably slows the hand-off)
        if (__builtin_expect((spins & 1023u) == 1023u, 0)) {
            // (made wave-uniform by hand: a per-lane load would make this exit, and with it
            // every value live across the chain loop, divergent in the compiler's eyes)
            const int ab = __builtin_amdgcn_readfirstlane(
                (int)(__hip_atomic_load(&ctl[CTL_ABORT], __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT) != 0));
            if (ab != 0 || spins > (1u << 23)) {
                // every lane stores the same word: a lane-0 branch right at this exit would
                // again make the exit look divergent
                if (ab == 0)
                    __hip_atomic_store(&ctl[CTL_ABORT], 1LL, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
#pragma unroll
        for (int q = 0; q < NVAL; ++q)
            if (pend[q])
                val[q] = load_published<SYS>(addr[q]);
    }
}

// The same wait with K polls kept in flight (in-order return: the oldest is checked while
// the younger ones travel), so that memory is sampled every round trip / K.
template <int K, int SYS, class Idle = NoIdle>
__device__ __forceinline__ bool poll_ring(const double *const (&addr)[2], double (&v)[K][2],
                                          double (&val)[2], long long *ctl,
                                          unsigned long long *nspin = nullptr, Idle idle = Idle())
{
    for (unsigned spins = 0;; ++spins) {
        idle();
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const bool pend0 = __any(mc3_bits(v[q][0]) == ROW_PENDING);
            const bool pend1 = __any(mc3_bits(v[q][1]) == ROW_PENDING);
            if (!pend0 && !pend1) {
                val[0] = v[q][0];
                val[1] = v[q][1];
                return true;
            }
            if (nspin)
                *nspin += 1;
            // only what is still missing is asked for again: every extra request on the
            // fabric slows everybody's hand-off (K = 2, 3, 4 polls in flight: 181, 173,
            // 140 M chain-steps/s against 197 for one)
            // (asked again at once: a pause of 64 / 128 / 256 / 512 cycles here measured
            // 197 / 195 / 191 / 182 M chain-steps/s against 205 without)
            if (K > 1 || pend0)
                v[q][0] = load_published<SYS>(addr[0]);
            if (K > 1 || pend1)
                v[q][1] = load_published<SYS>(addr[1]);
        }
        if (__builtin_expect((spins & 255u) == 255u, 0)) {
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
    }
}

// Keep a wave-uniform value where it was computed: an empty asm that "uses" it, so
// neither the IR-level sinking pass nor the scheduler can move the (scalar) Philox
// arithmetic of the next iteration out of the shadow of the partner-row loads.
__device__ __forceinline__ void pin_scalar(uint64_t &q)
{
    asm volatile("; pin" : "+s"(q));
}

// the wave-uniform draws of one iteration (slots 0..2)
struct Draws {
    Draw2 d0, d1, d2;
};

__device__ __forceinline__ Draws make_draws(const DevProblem &P, uint64_t iter, int c, int sampler)
{
    Draws D;
    D.d0 = draw2(P.seed, iter, (uint32_t)c, 0u);
    D.d1 = D.d0;
    D.d2 = D.d0;
    if (sampler == MC3HIP_SNOOKER) {
        D.d1 = draw2(P.seed, iter, (uint32_t)c, 1u);
        D.d2 = draw2(P.seed, iter, (uint32_t)c, 2u);
    }
    return D;
}

// The same draws for 21 consecutive iterations at once, on the vector unit: lane
// 3*(iter - base) + slot holds the four words of (iter, chain, slot).  The scalar Philox
// of make_draws costs a wave ~350 issue slots per call; it is free only while the wave
// waits for memory anyway.  Where nothing is left to wait for (persistent snooker, whose
// rows are requested an iteration ahead) a batch costs ~50 cycles per iteration instead.
struct DrawBatch {
    uint32_t w0, w1, w2, w3;
    long long base;
};
constexpr int DRAW_BATCH = 21;

__device__ __forceinline__ void fill_draws(DrawBatch &B, const DevProblem &P, long long iter0,
                                           int c)
{
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t it = (uint64_t)iter0 + (uint64_t)(lane / 3u);
    uint32_t w[4];
    philox4x32_10((uint32_t)it, (uint32_t)(it >> 32), (uint32_t)c, lane % 3u, (uint32_t)P.seed,
                  (uint32_t)(P.seed >> 32), w);
    B.w0 = w[0];
    B.w1 = w[1];
    B.w2 = w[2];
    B.w3 = w[3];
    B.base = iter0;
}

__device__ __forceinline__ Draw2 batch_draw(const DrawBatch &B, int l)
{
    Draw2 d;
    d.q0 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)B.w0, l) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)B.w1, l) << 32);
    d.q1 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)B.w2, l) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)B.w3, l) << 32);
    return d;
}

__device__ __forceinline__ Draws get_draws(DrawBatch &B, const DevProblem &P, long long iter,
                                           int c)
{
    if (iter < B.base || iter >= B.base + DRAW_BATCH)
        fill_draws(B, P, iter, c);
    const int l = 3 * (int)(iter - B.base);
    Draws D;
    D.d0 = batch_draw(B, l);
    D.d1 = batch_draw(B, l + 1);
    D.d2 = batch_draw(B, l + 2);
    return D;
}

// coefficient j of a fully-polynomial parameter vector: the free parameter it is (kf[j]) or
// its fixed value (fx[j], kf[j] = -1); wave-uniform, computed once per launch
template <int MT>
__device__ __forceinline__ void coefficient_map(const DevProblem &P, int (&kf)[MT], double (&fx)[MT])
{
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        kf[j] = -1;
        fx[j] = P.params[j];
    }
    for (int k = 0; k < P.nfree; ++k) {
        const int j = P.ifree[k];
#pragma unroll
        for (int q = 0; q < MT; ++q)
            kf[q] = (q == j) ? k : kf[q];
    }
    // shared parameters copy their source, in order (chain.py:246-247)
    for (int sh = 0; sh < P.nshare; ++sh) {
        const int j = P.ishare[sh], src = P.share_src[sh];
        int k = -1;
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < MT; ++q) {
            k = (q == src) ? kf[q] : k;
            v = (q == src) ? fx[q] : v;
        }
#pragma unroll
        for (int q = 0; q < MT; ++q) {
            kf[q] = (q == j) ? k : kf[q];
            fx[q] = (q == j) ? v : fx[q];
        }
    }
}

// prior_term for the lean FX instances: the proposal is in the lanes (free parameter k in
// lane k), not in LDS; same terms in the same order
template <int MT>
__device__ __forceinline__ double prior_from_lanes(const DevProblem &P, double npk,
                                                   const int (&kf)[MT], const double (&fx)[MT])
{
    double pr = 0.0;
    for (int g = 0; g < P.ngprior; ++g) {
        const int j = P.igprior[g];
        int k = -1;
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < MT; ++q) {
            k = (q == j) ? kf[q] : k;
            v = (q == j) ? fx[q] : v;
        }
        if (k >= 0)
            v = make_double(__builtin_amdgcn_readlane(__double2loint(npk), k),
                            __builtin_amdgcn_readlane(__double2hiint(npk), k));
        const double d = v - P.prior[j];
        const double s = (d > 0.0) ? d / P.priorup[j] : d / P.priorlow[j];
        pr = pr + s * s;
    }
    return pr;
}

// The same terms with the lanes doing the work: prior_from_lanes reads the priors' indices,
// centres and widths from global memory every time -- four dependent scalar loads per prior,
// ~0.3 us of latency per prior in a loop that has nothing else to wait for (off_prior: 137 M
// chain-steps/s against 215 without priors).  Here lane g keeps prior g (index of its free
// parameter or its fixed value, centre, widths) in registers for the whole launch; an iteration
// gathers the proposed values through the crossbar, every lane divides at once, and the
// squares are added in the priors' order (lane_serial_sum): the same operations on the same
// values in the same order -- bit-identical.
template <int N>
struct PriorLanes {
    int src, n;
    double fixedv, centre, lo, up;
    __device__ __forceinline__ void init(const DevProblem &P, const int (&kf)[N],
                                         const double (&fx)[N], int lane)
    {
        n = P.ngprior;
        src = -1;
        fixedv = 0.0;
        centre = 0.0;
        lo = up = 1.0;
        if (lane < n) {
            const int j = P.igprior[lane];
            int k = -1;
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < N; ++q) {
                k = (q == j) ? kf[q] : k;
                v = (q == j) ? fx[q] : v;
            }
            src = k;
            fixedv = v;
            centre = P.prior[j];
            lo = P.priorlow[j];
            up = P.priorup[j];
        }
    }
    __device__ __forceinline__ double term(double npk, int lane) const
    {
        const double got = lane_gather(npk, src < 0 ? lane : src);
        const double v = src >= 0 ? got : fixedv;
        const double d = v - centre;
        const double s = (d > 0.0) ? d / up : d / lo;
        return lane_serial_sum(s * s, n);
    }
};

// The Metropolis test exp(t) > u (chain.py:257-260, t = (chisq_cur - chisq_new)/2, no
// Hastings factor) without the exponential on the critical path.  u is known long before t
// -- it is drawn from (seed, iteration, chain) alone -- so L = log(u) is computed in the
// shadow of the wait for the partner rows, and the dependent chain chisq -> exp -> compare
// -> publish (a single wave has nothing to hide it behind) loses its ~25 serial FP64 steps:
//   t >= L + d  =>  accept,   t <= L - d  =>  reject,   d = 4e-15 (1 + |L|),
// and ONLY inside that band (a handful of ulps wide: practically never) the exponential
// itself decides.  The outcome is that of mc3_det_exp(t) > u in every case, so results stay
// bit-identical to the checker, which always evaluates the exponential:
//   mc3_det_exp and mc3_det_log are within 2 ulp of exp and log (tests/test_detmath_philox.py),
//   i.e. det_exp(t) = exp(t)(1 + e1), |e1| < 5e-16, and |L - log u| < 5e-16 |L|; for
//   t >= L + d:  det_exp(t) >= u exp(d - 5e-16 |L|)(1 - 5e-16) >= u exp(3e-15)(1 - 5e-16) > u,
//   and symmetrically below.  u = 0 (log = -inf) and a NaN t take the exact path.
__device__ __forceinline__ void accept_window(double u, double &lo, double &hi)
{
    const double L = mc3_det_log(u);
    const double d = 4e-15 * (1.0 + __builtin_fabs(L));
    // (u == 0: an empty window that nothing falls outside of -- lo = +inf, hi = -inf would
    // decide everything; NaNs decide nothing)
    const bool usable = u > 0.0;
    lo = usable ? L - d : mc3_from_bits(0x7ff8000000000000ull);
    hi = usable ? L + d : mc3_from_bits(0x7ff8000000000000ull);
}

__device__ __forceinline__ bool metropolis_accept(double t, double u, double lo, double hi)
{
    const bool sure_yes = t >= hi, sure_no = t <= lo;        // (false for NaNs)
    if (__builtin_expect(sure_yes || sure_no, 1))
        return sure_yes;
    return mc3_det_exp(t) > u;
}

// DEMC partners, chain.py:221-232: r1 != c, r2 != c, r2 != r1, from the two 32-bit
// halves of one draw
__device__ __forceinline__ void demc_partners(uint64_t q0, int n, int c, int &r1, int &r2)
{
    r1 = 1 + (int)bounded32((uint32_t)(q0 >> 32), (uint32_t)(n - 1));
    if (r1 == c)
        r1 = 0;
    r2 = r1 + 2 + (int)bounded32((uint32_t)q0, (uint32_t)(n - 2));   // < 2n
    r2 = (r2 >= n) ? r2 - n : r2;                                      // (...) % n
    if (r2 == c)
        r2 = (r1 + 1 == n) ? 0 : r1 + 1;                               // (r1+1) % n
}

}  // namespace mc3hip
