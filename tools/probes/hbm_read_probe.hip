// hbm_read_probe.hip -- the denominator of the contractual roofline, MEASURED: how many bytes
// per second this very box streams out of HBM into registers ("measure
// it on the box with a streaming-read probe and use the measured value").
//
// A grid of 256 CUs x 8 workgroups of 256 threads sweeps a buffer far beyond the 256 MiB
// Infinity Cache (default 4 GiB) with 16-byte loads, 4 in flight per thread, non-temporal so
// that nothing of one sweep is left in a cache for the next; every loaded word is folded into
// a checksum (the loads cannot be optimised away).  Timed with HIP events over `reps` sweeps
// after one warm-up sweep; prints ONE JSON line.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/_bin/hbm_read_probe tools/probes/hbm_read_probe.hip
//   tools/probes/_bin/hbm_read_probe [GiB [reps [device]]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(expr)                                                                  \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s failed: %s (line %d)\n", #expr, hipGetErrorString(e_), \
                    __LINE__);                                                       \
            exit(2);                                                                 \
        }                                                                            \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// every thread reads UNROLL x 16 bytes per trip, the workgroups of the grid interleaved at the
// granularity of one trip (256 threads x 16 B x UNROLL = 16 KiB): consecutive workgroups read
// consecutive 16 KiB, so all HBM channels are busy at any moment
template <int UNROLL>
__global__ __launch_bounds__(256) void sweep(const u32x4 *__restrict__ buf, size_t nvec,
                                             uint32_t *__restrict__ sink)
{
    const size_t trip = (size_t)256 * UNROLL;
    const size_t stride = (size_t)gridDim.x * trip;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (size_t base = (size_t)blockIdx.x * trip + threadIdx.x; base + (UNROLL - 1) * 256 < nvec;
         base += stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
            v[k] = __builtin_nontemporal_load(buf + base + (size_t)k * 256);
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
            acc ^= v[k];
    }
    const uint32_t x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x9E3779B9u)            // (never true for the pattern below: keeps the loads alive)
        sink[0] = x;
}

__global__ void fill(u32x4 *buf, size_t nvec)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t w = (uint32_t)(i * 2654435761u) | 1u;
        buf[i] = u32x4{w, w + 2u, w + 4u, w + 6u};
    }
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const int reps = argc > 2 ? atoi(argv[2]) : 8;
    const int device = argc > 3 ? atoi(argv[3]) : 0;
    CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, device));
    const size_t bytes = (size_t)(gib * (double)(1ull << 30)) & ~(size_t)0xFFFF;
    const size_t nvec = bytes / 16;
    u32x4 *buf;
    uint32_t *sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 4));
    fill<<<prop.multiProcessorCount * 8, 256>>>(buf, nvec);
    CHECK(hipDeviceSynchronize());
    const int grid = prop.multiProcessorCount * 8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    double best = 0.0, sum = 0.0;
    sweep<4><<<grid, 256>>>(buf, nvec, sink);          // warm-up (clocks, page tables)
    CHECK(hipDeviceSynchronize());
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        sweep<4><<<grid, 256>>>(buf, nvec, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double gbs = (double)bytes / (ms * 1e-3) / 1e9;
        sum += gbs;
        if (gbs > best)
            best = gbs;
    }
    printf("{\"probe\": \"hbm_streaming_read\", \"device\": \"%s\", \"cus\": %d, \"bytes\": %zu, "
           "\"reps\": %d, \"GBps_mean\": %.1f, \"GBps_best\": %.1f, "
           "\"shape\": \"nontemporal 16-byte loads, 4 in flight per thread, 8 workgroups of 256 "
           "threads per CU, buffer beyond the Infinity Cache\"}\n",
           prop.name, prop.multiProcessorCount, bytes, reps, sum / reps, best);
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
