// xgmi_handoff.hip -- the one number DESIGN.md section 7 calls open: what does ONE hop of the
// in-launch hand-off cost when producer and consumer sit on DIFFERENT GPUs?
//
// The product's protocol, reduced to its core (handoff.h, run_chain.h P2P): a producer
// publishes an 8-byte granule with a system-scope store INTO THE CONSUMER'S memory
// (fine-grained allocation, peer-to-peer over xGMI); the consumer polls its LOCAL copy
// with system-scope loads until the granule is there.  Here two single-wave kernels, one
// per device, play ping-pong with sequence numbers: every round trip is two such hops.
// Reported: microseconds per hop, for
//   (a) two devices (A -> B -> A), the cross-GPU hop;
//   (b) the same pair of kernels on ONE device (two CUs, fine-grained memory, system
//       scope): what the peer-to-peer flavour of the kernels pays even without a link;
//   (c) one device, ordinary memory, agent scope, leader and follower two workgroups of ONE
//       launch: the single-GPU hand-off of the persistent kernels (MI355X_MICROARCH.md
//       "handoff-1to1").  (Two separate launches do not see each other's agent-scope
//       stores to ordinary memory on this stack: measured, both gave up.)
// Every wait is bounded, so a missing peer mapping ends in an error message, not a hang.
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/xgmi_handoff tools/probes/xgmi_handoff.hip
//   /tmp/xgmi_handoff [devA devB [round trips]]     (default: 0 1; one GPU: (b) and (c) only)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(expr)                                                                  \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s failed: %s (line %d)\n", #expr, hipGetErrorString(e_), \
                    __LINE__);                                                       \
            exit(2);                                                                 \
        }                                                                            \
    } while (0)

template <bool SYS>
__device__ __forceinline__ unsigned long long load_granule(const unsigned long long *p)
{
    if constexpr (SYS)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool SYS>
__device__ __forceinline__ void store_granule(unsigned long long *p, unsigned long long v)
{
    if constexpr (SYS)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// mine: polled here, written by the peer; theirs: the peer's copy, written from here.
// status[0] = rounds completed, status[1] = 1 when a wait gave up.
template <bool SYS>
__global__ __launch_bounds__(64) void pingpong(unsigned long long *mine, unsigned long long *theirs,
                                               int rounds, int first, unsigned long long *status)
{
    int done = 0;
    for (int i = 1; i <= rounds; ++i) {
        if (first)
            store_granule<SYS>(theirs, (unsigned long long)i);
        unsigned spins = 0;
        while (load_granule<SYS>(mine) < (unsigned long long)i) {
            if (++spins > (1u << 24)) {
                if (threadIdx.x == 0) {
                    status[0] = done;
                    status[1] = 1;
                }
                return;
            }
        }
        if (!first)
            store_granule<SYS>(theirs, (unsigned long long)i);
        done = i;
    }
    if (threadIdx.x == 0) {
        status[0] = done;
        status[1] = 0;
    }
}

// (c): leader and follower as workgroups 0 and 1 of ONE launch (consecutive workgroups go to
// different XCDs), exactly the situation of the persistent kernels on one GPU
__global__ __launch_bounds__(64) void pingpong_one_launch(unsigned long long *flag_a,
                                                          unsigned long long *flag_b, int rounds,
                                                          unsigned long long *status)
{
    const int first = blockIdx.x == 0;
    unsigned long long *mine = first ? flag_a : flag_b, *theirs = first ? flag_b : flag_a;
    int done = 0;
    for (int i = 1; i <= rounds; ++i) {
        if (first)
            store_granule<false>(theirs, (unsigned long long)i);
        unsigned spins = 0;
        while (load_granule<false>(mine) < (unsigned long long)i) {
            if (++spins > (1u << 22)) {
                if (threadIdx.x == 0 && first) {
                    status[0] = done;
                    status[1] = 1;
                }
                return;
            }
        }
        if (!first)
            store_granule<false>(theirs, (unsigned long long)i);
        done = i;
    }
    if (threadIdx.x == 0 && first) {
        status[0] = done;
        status[1] = 0;
    }
}

struct Side {
    int dev;
    hipStream_t stream;
    unsigned long long *flag, *status;
};

static Side make_side(int dev, bool fine)
{
    Side s;
    s.dev = dev;
    CHECK(hipSetDevice(dev));
    CHECK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    void *p = nullptr;
    if (fine)
        CHECK(hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained));
    else
        CHECK(hipMalloc(&p, 4096));
    CHECK(hipMemset(p, 0, 4096));
    s.flag = static_cast<unsigned long long *>(p);
    CHECK(hipMalloc(&p, 64));
    CHECK(hipMemset(p, 0, 64));
    s.status = static_cast<unsigned long long *>(p);
    CHECK(hipDeviceSynchronize());
    return s;
}

template <bool SYS>
static double run(Side &a, Side &b, int rounds)
{
    CHECK(hipSetDevice(a.dev));
    CHECK(hipMemset(a.flag, 0, 8));
    CHECK(hipDeviceSynchronize());
    CHECK(hipSetDevice(b.dev));
    CHECK(hipMemset(b.flag, 0, 8));
    CHECK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    // the follower first: it only waits
    CHECK(hipSetDevice(b.dev));
    hipLaunchKernelGGL(pingpong<SYS>, dim3(1), dim3(64), 0, b.stream, b.flag, a.flag, rounds, 0,
                       b.status);
    CHECK(hipSetDevice(a.dev));
    hipLaunchKernelGGL(pingpong<SYS>, dim3(1), dim3(64), 0, a.stream, a.flag, b.flag, rounds, 1,
                       a.status);
    CHECK(hipStreamSynchronize(a.stream));
    CHECK(hipSetDevice(b.dev));
    CHECK(hipStreamSynchronize(b.stream));
    const double us =
        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long st[2] = {0, 0};
    CHECK(hipSetDevice(a.dev));
    CHECK(hipMemcpy(st, a.status, sizeof st, hipMemcpyDeviceToHost));
    if (st[1] != 0 || (int)st[0] != rounds) {
        fprintf(stderr, "  gave up after %llu of %d round trips (no peer mapping?)\n", st[0], rounds);
        return -1.0;
    }
    return us / (2.0 * rounds);
}

int main(int argc, char **argv)
{
    int ndev = 0;
    CHECK(hipGetDeviceCount(&ndev));
    int da = argc > 1 ? atoi(argv[1]) : 0;
    int db = argc > 2 ? atoi(argv[2]) : (ndev > 1 ? 1 : 0);
    const int rounds = argc > 3 ? atoi(argv[3]) : 200000;
    if (da >= ndev || db >= ndev) {
        fprintf(stderr, "devices %d, %d: only %d present\n", da, db, ndev);
        return 2;
    }
    printf("{\"devices_present\": %d, \"round_trips\": %d", ndev, rounds);
    if (da != db) {
        int ab = 0, ba = 0;
        CHECK(hipDeviceCanAccessPeer(&ab, da, db));
        CHECK(hipDeviceCanAccessPeer(&ba, db, da));
        if (!ab || !ba) {
            fprintf(stderr, "devices %d and %d cannot access each other\n", da, db);
            return 2;
        }
        CHECK(hipSetDevice(da));
        (void)hipDeviceEnablePeerAccess(db, 0);
        CHECK(hipSetDevice(db));
        (void)hipDeviceEnablePeerAccess(da, 0);
        (void)hipGetLastError();
        Side a = make_side(da, true), b = make_side(db, true);
        run<true>(a, b, 2000);                                    // warm-up
        const double hop = run<true>(a, b, rounds);
        printf(", \"cross_gpu\": {\"devices\": [%d, %d], \"us_per_hop\": %.3f}", da, db, hop);
    }
    {
        Side a = make_side(da, true), b = make_side(da, true);
        run<true>(a, b, 2000);
        printf(", \"one_gpu_finegrained_system_scope\": {\"device\": %d, \"us_per_hop\": %.3f}", da,
               run<true>(a, b, rounds));
    }
    {
        Side a = make_side(da, false), b = make_side(da, false);
        auto once = [&](int n) {
            CHECK(hipSetDevice(da));
            CHECK(hipMemset(a.flag, 0, 8));
            CHECK(hipMemset(b.flag, 0, 8));
            CHECK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(pingpong_one_launch, dim3(2), dim3(64), 0, a.stream, a.flag, b.flag,
                               n, a.status);
            CHECK(hipStreamSynchronize(a.stream));
            const double us = std::chrono::duration<double, std::micro>(
                                  std::chrono::steady_clock::now() - t0).count();
            unsigned long long st[2] = {0, 0};
            CHECK(hipMemcpy(st, a.status, sizeof st, hipMemcpyDeviceToHost));
            return (st[1] != 0 || (int)st[0] != n) ? -1.0 : us / (2.0 * n);
        };
        once(2000);
        printf(", \"one_gpu_agent_scope_one_launch\": {\"device\": %d, \"us_per_hop\": %.3f}", da,
               once(rounds));
    }
    printf("}\n");
    return 0;
}
