// Probe: v_mfma_f64_16x16x4_f64 on gfx950 -- (1) is D = A*B + C over k = 0..3 the same
// number as the sequential chain fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,c))))?
// (2) how many cycles does one instruction occupy a SIMD's matrix pipe?
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f64_probe mfma_f64_probe.hip && ./mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void mfma_once(const double *A, const double *B, const double *C, double *D)
{
    // A [16][4] row-major (row i, k), B [4][16] (k, col j), C/D [16][16]
    const int lane = threadIdx.x, ij = lane & 15, k = lane >> 4;
    const double a = A[ij * 4 + k], b = B[k * 16 + ij];
    v4d c;
    for (int r = 0; r < 4; ++r)
        c[r] = C[(k + 4 * r) * 16 + ij];          // row = (lane>>4) + 4 r, col = lane&15
    v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r)
        D[(k + 4 * r) * 16 + ij] = d[r];
}

__global__ void mfma_rate(double *out, int n, unsigned long long *cycles)
{
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0 && blockIdx.x == 0)
        cycles[0] = t1 - t0;
}

__global__ void fma_rate(double *out, int n, unsigned long long *cycles)
{
    double a = 1.0 + threadIdx.x * 1e-9, x0 = 0, x1 = 0, x2 = 0, x3 = 0, x4 = 0, x5 = 0, x6 = 0, x7 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        x0 = __builtin_fma(a, x0, a); x1 = __builtin_fma(a, x1, a); x2 = __builtin_fma(a, x2, a);
        x3 = __builtin_fma(a, x3, a); x4 = __builtin_fma(a, x4, a); x5 = __builtin_fma(a, x5, a);
        x6 = __builtin_fma(a, x6, a); x7 = __builtin_fma(a, x7, a);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        cycles[1] = t1 - t0;
}

int main()
{
    std::vector<double> A(64), B(64), C(256), D(256);
    double *dA, *dB, *dC, *dD;
    unsigned long long *dcy, cy[2] = {0, 0};
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 256 * 8); hipMalloc(&dD, 8 << 20);
    hipMalloc(&dcy, 16);
    srand(7);
    int same_fwd = 0, same_rev = 0, same_pair = 0, total = 0;
    for (int trial = 0; trial < 200; ++trial) {
        for (auto &v : A) v = (rand() / (double)RAND_MAX - 0.5) * std::ldexp(1.0, rand() % 20 - 10);
        for (auto &v : B) v = (rand() / (double)RAND_MAX - 0.5) * std::ldexp(1.0, rand() % 20 - 10);
        for (auto &v : C) v = (trial % 2) ? 0.0 : (rand() / (double)RAND_MAX - 0.5);
        hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 256 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double f = C[i * 16 + j], r = C[i * 16 + j];
                for (int k = 0; k < 4; ++k)
                    f = std::fma(A[i * 4 + k], B[k * 16 + j], f);
                for (int k = 3; k >= 0; --k)
                    r = std::fma(A[i * 4 + k], B[k * 16 + j], r);
                const double p = std::fma(A[i*4+2], B[2*16+j], A[i*4+3]*B[3*16+j]) +
                                 std::fma(A[i*4+0], B[0*16+j], A[i*4+1]*B[1*16+j]) + C[i*16+j];
                const double got = D[i * 16 + j];
                same_fwd += (got == f);
                same_rev += (got == r);
                same_pair += (got == p);
                total += 1;
            }
    }
    printf("mfma_f64_16x16x4: equal to sequential fma chain k=0..3: %d / %d; k=3..0: %d; pairwise: %d\n",
           same_fwd, total, same_rev, same_pair);
    const int n = 20000;
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    for (int waves = 1; waves <= 4; waves *= 2) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_rate, dim3(256 * 4), dim3(64 * waves), 0, 0, dD, n, dcy);
        hipEventRecord(e1, 0);
        hipLaunchKernelGGL(fma_rate, dim3(256 * 4), dim3(64 * waves), 0, 0, dD, n, dcy);
        hipEventRecord(e2, 0);
        hipDeviceSynchronize();
        float ms0 = 0, ms1 = 0;
        hipEventElapsedTime(&ms0, e0, e1); hipEventElapsedTime(&ms1, e1, e2);
        printf("   wall: MFMA %.1f TFLOP/s f64 (%.2f ms), vector FMA %.1f TFLOP/s (%.2f ms)\n",
               1024.0 * waves * n * 4 * 2048.0 / (ms0 * 1e-3) / 1e12, ms0,
               1024.0 * waves * n * 8 * 128.0 / (ms1 * 1e-3) / 1e12, ms1);
        hipMemcpy(cy, dcy, 16, hipMemcpyDeviceToHost);
        // blocks of `waves` waves, 4 blocks per CU -> waves per SIMD = waves
        printf("%d wave(s)/SIMD: %.1f cycles per MFMA f64 16x16x4 per wave (1024 FMA); "
               "%.2f cycles per v_fma_f64 per wave (64 FMA)\n", waves,
               (double)cy[0] / (4.0 * n), (double)cy[1] / (8.0 * n));
    }
    return 0;
}
