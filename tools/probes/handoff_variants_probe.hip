// handoff_variants_probe -- one hop of the in-launch hand-off (two single-wave workgroups of ONE
// launch playing ping-pong with sequence numbers, as xgmi_handoff.hip (c)), by HOW the word is
// stored and HOW it is polled.  The product stores with `global_store_dwordx2 sc1` and polls with
// `global_load_dwordx2 sc1` (0.53 us per hop, profiles/r03_handoff_probe_one_gpu.json); is any
// other pair of instructions faster between two CUs of this chip?
//   loads:  0 vector sc1 (the product)   1 vector sc0 sc1   2 scalar s_load glc
//           3 scalar s_dcache_inv + s_load   4 returning atomic OR of 0 (sc0 sc1)
//   stores: 0 vector sc1 (the product)   1 vector sc0 sc1   2 atomic swap (no return)
//           3 plain store + buffer_wbl2 sc1
// Every wait is bounded.   handoff_variants_probe [round trips = 100000] [far = 1]
// (far = 1: workgroups 0 and 1 -- different chiplets under round-robin placement; far = 0:
// workgroups 0 and 8 -- the same chiplet)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

template <int L>
__device__ __forceinline__ unsigned long long poll(const unsigned long long *p)
{
    unsigned long long v;
    if constexpr (L == 0) {
        asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=v"(v) : "v"(p) : "memory");
    } else if constexpr (L == 1) {
        asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=v"(v) : "v"(p) : "memory");
    } else if constexpr (L == 2) {
        asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)"
                     : "=s"(v) : "s"(p) : "memory");
    } else if constexpr (L == 3) {
        asm volatile("s_dcache_inv\n\ts_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=s"(v) : "s"(p) : "memory");
    } else {
        asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=v"(v) : "v"(p), "v"(0ull) : "memory");
    }
    return v;
}

template <int S>
__device__ __forceinline__ void publish(unsigned long long *p, unsigned long long v)
{
    if constexpr (S == 0) {
        asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    } else if constexpr (S == 1) {
        asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    } else if constexpr (S == 2) {
        asm volatile("global_atomic_swap_x2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    } else {
        asm volatile("global_store_dwordx2 %0, %1, off\n\tbuffer_wbl2 sc1"
                     :: "v"(p), "v"(v) : "memory");
    }
}

template <int L, int S>
__global__ __launch_bounds__(64) void pingpong(unsigned long long *flag_a, unsigned long long *flag_b,
                                               int rounds, int partner, unsigned long long *status)
{
    if (blockIdx.x != 0 && (int)blockIdx.x != partner)
        return;
    const int first = blockIdx.x == 0;
    unsigned long long *mine = first ? flag_a : flag_b, *theirs = first ? flag_b : flag_a;
    int done = 0;
    for (int i = 1; i <= rounds; ++i) {
        if (first && threadIdx.x == 0)
            publish<S>(theirs, (unsigned long long)i);
        unsigned spins = 0;
        for (;;) {
            unsigned long long v = poll<L>(mine);
            v = __builtin_amdgcn_readfirstlane((unsigned)v) |
                ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32);
            if (v >= (unsigned long long)i)
                break;
            if (++spins > (1u << 22)) {
                if (threadIdx.x == 0 && first) {
                    status[0] = done;
                    status[1] = 1;
                }
                return;
            }
        }
        if (!first && threadIdx.x == 0)
            publish<S>(theirs, (unsigned long long)i);
        done = i;
    }
    if (threadIdx.x == 0 && first) {
        status[0] = done;
        status[1] = 0;
    }
}

#define CHECK(expr)                                                                    \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s failed: %s\n", #expr, hipGetErrorString(e_));          \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

template <int L, int S>
static double run(unsigned long long *flags, unsigned long long *status, int rounds, int partner)
{
    CHECK(hipMemset(flags, 0, 8192));
    CHECK(hipMemset(status, 0, 64));
    CHECK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL((pingpong<L, S>), dim3(16), dim3(64), 0, 0, flags, flags + 512, rounds,
                       partner, status);
    CHECK(hipDeviceSynchronize());
    const double us =
        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long st[2] = {0, 0};
    CHECK(hipMemcpy(st, status, sizeof st, hipMemcpyDeviceToHost));
    if (st[1] != 0 || (int)st[0] != rounds)
        return -1.0;                       // (the word never arrived: this pair does not work)
    return us / (2.0 * rounds);
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 100000;
    const int partner = (argc > 2 && atoi(argv[2]) == 0) ? 8 : 1;
    unsigned long long *flags = nullptr, *status = nullptr;
    CHECK(hipMalloc(&flags, 8192));
    CHECK(hipMalloc(&status, 64));
    const char *ln[] = {"vector sc1", "vector sc0 sc1", "scalar glc", "scalar inv+load",
                        "atomic or 0"};
    const char *sn[] = {"store sc1", "store sc0 sc1", "atomic swap", "plain + wbl2"};
    printf("# us per hop, workgroups 0 and %d of one launch, %d round trips (-1: never arrived)\n",
           partner, rounds);
#define ROW(L)                                                                              \
    printf("%-16s", ln[L]);                                                                 \
    printf(" %8.3f", run<L, 0>(flags, status, rounds, partner));                           \
    printf(" %8.3f", run<L, 1>(flags, status, rounds, partner));                           \
    printf(" %8.3f", run<L, 2>(flags, status, rounds, partner));                           \
    printf(" %8.3f\n", run<L, 3>(flags, status, rounds, partner));
    printf("%-16s %8s %8s %8s %8s\n", "load \\ store", sn[0], "sc0 sc1", sn[2], sn[3]);
    run<0, 0>(flags, status, 2000, partner);
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4)
    return 0;
}
