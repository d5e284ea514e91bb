// Probe: how is v_mfma_f64_16x16x4_f64 issued on gfx950?  (1) rate of one wave per SIMD
// against the number of independent accumulators (a dependent-latency or an issue limit?);
// (2) do vector FP64 fused multiply-adds of the same wave run in the shadow of the matrix
// instruction (separate pipes) or do they share it?
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f64_issue_probe mfma_f64_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

template <int NA, int NV>
__global__ void mix(double *out, int n)
{
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    v4d c[NA > 0 ? NA : 1];
    double x[NV > 0 ? NV : 1];
    for (int i = 0; i < (NA > 0 ? NA : 1); ++i) c[i] = v4d{0, 0, 0, 0};
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) x[i] = lane * 1e-9 * i;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            x[i] = __builtin_fma(a, x[i], b);
        // NA matrix instructions, each followed by NV / NA vector ones
        if constexpr (NA > 0 && NV > 0) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NV / NA, 0);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < (NA > 0 ? NA : 1); ++i) s += c[i][0] + c[i][3];
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NA, int NV>
void run(double *d, int waves)
{
    const int n = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mix<NA, NV>), dim3(256 * 4), dim3(64 * waves), 0, 0, d, 100);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mix<NA, NV>), dim3(256 * 4), dim3(64 * waves), 0, 0, d, n);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_it_ns = ms * 1e6 / n;            // one iteration of one wave... of all waves/SIMD
    printf("  %d wave(s)/SIMD, %2d matrix + %2d vector per iteration: %7.1f ns per iteration"
           " = %6.1f cycles at 2.4 GHz;  matrix %5.1f TFLOP/s, vector %5.1f TFLOP/s\n",
           waves, NA, NV, per_it_ns, per_it_ns * 2.4,
           1024.0 * waves * n * NA * 2048.0 / (ms * 1e-3) / 1e12,
           1024.0 * waves * n * NV * 128.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    double *d;
    hipMalloc(&d, 64 << 20);
    for (int waves = 1; waves <= 2; ++waves) {
        printf("independent accumulators:\n");
        run<1, 0>(d, waves); run<2, 0>(d, waves); run<4, 0>(d, waves); run<8, 0>(d, waves);
        printf("vector only:\n");
        run<0, 16>(d, waves); run<0, 32>(d, waves);
        printf("both, interleaved:\n");
        run<4, 16>(d, waves); run<4, 32>(d, waves); run<4, 64>(d, waves); run<8, 64>(d, waves);
    }
    return 0;
}
