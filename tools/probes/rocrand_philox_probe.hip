// rocrand_philox_probe.hip -- the claim of mc3_amd/csrc/philox.h:7-9 put to the test against
// rocRAND ITSELF: "the counter/key layout equals rocRAND's rocrand_state_philox4x32_10 after
// rocrand_init(seed, subsequence = chain | slot << 32, offset = 4*iteration)".
//
// Reads (seed, n, then n x (iteration, chain, slot)) as 64-bit little-endian words from stdin,
// draws rocrand4() from a state initialised exactly so on the device, writes n x 4 uint32 to
// stdout.  tests/test_gpu_api.py compares them word for word with the library's own rounds
// (mc3hip_selftest_philox) and with the checker's.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/_bin/rocrand_philox_probe tools/probes/rocrand_philox_probe.hip
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(expr)                                                                  \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s failed: %s (line %d)\n", #expr, hipGetErrorString(e_), \
                    __LINE__);                                                       \
            exit(2);                                                                 \
        }                                                                            \
    } while (0)

__global__ void draw(unsigned long long seed, const unsigned long long *req, int n, uint32_t *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const unsigned long long iteration = req[3 * i], chain = req[3 * i + 1], slot = req[3 * i + 2];
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, chain | (slot << 32), 4ull * iteration, &st);
    const uint4 w = rocrand4(&st);
    out[4 * i + 0] = w.x;
    out[4 * i + 1] = w.y;
    out[4 * i + 2] = w.z;
    out[4 * i + 3] = w.w;
}

int main()
{
    unsigned long long head[2];
    if (fread(head, 8, 2, stdin) != 2)
        return 3;
    const unsigned long long seed = head[0];
    const int n = (int)head[1];
    std::vector<unsigned long long> req(3 * (size_t)n);
    if (fread(req.data(), 8, req.size(), stdin) != req.size())
        return 3;
    unsigned long long *dreq;
    uint32_t *dout;
    CHECK(hipMalloc(&dreq, req.size() * 8));
    CHECK(hipMalloc(&dout, (size_t)n * 16));
    CHECK(hipMemcpy(dreq, req.data(), req.size() * 8, hipMemcpyHostToDevice));
    draw<<<(n + 63) / 64, 64>>>(seed, dreq, n, dout);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> out(4 * (size_t)n);
    CHECK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    fwrite(out.data(), 4, out.size(), stdout);
    return 0;
}
