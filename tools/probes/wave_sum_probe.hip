// Probe: the sum over a wave's 64 lanes as a chain of dependent results -- the product's six
// butterfly levels (lane swaps / DPP rotations + v_add_f64: core.h wave_sum) against the FP64
// matrix unit doing four of them in two instructions:
//   v_mfma_f64_16x16x4 with A = 1: D[i][j] = sum_k B[k][j], lane 16 k + j holds B[k][j]  -> every
//       lane l receives ((t[j] + t[16 + j]) + t[32 + j]) + t[48 + j], j = l mod 16
//   v_mfma_f64_4x4x4 (four blocks of sixteen lanes) with A = 1                          -> ... the
//       sum over lanes j, j + 4, j + 8, j + 12 of its row of sixteen
//   two DPP quad permutations + additions for the last four values.
// Prints, for one wave per SIMD on every CU: ns per sum (a dependent chain of sums), and whether
// the matrix variant's total equals the host's sum in the order claimed above, bit for bit.
// RESULT (profiles/r06_wave_sum_probe.txt): the butterfly 71 ns per sum and multiply-add -- under
// 10 ns per level --, the matrix variant 81 ns: not faster, so the operand layout of the 4x4x4
// instruction (which the comparison shows is NOT the one assumed here) was not pursued.
//   hipcc --offload-arch=gfx950 -O2 -o wave_sum_probe wave_sum_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double mk(int lo, int hi)
{
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double add_xor32(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return mk(a[0], b[0]) + mk(a[1], b[1]);
}
__device__ __forceinline__ double add_xor16(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return mk(a[0], b[0]) + mk(a[1], b[1]);
}
template <int CTRL>
__device__ __forceinline__ double add_dpp(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    const int hi2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return v + mk(lo2, hi2);
}
__device__ __forceinline__ double sum_butterfly(double t)
{
    t = add_xor32(t);
    t = add_xor16(t);
    t = add_dpp<0x128>(t);      // row_ror:8
    t = add_dpp<0x124>(t);      // row_ror:4
    t = add_dpp<0x122>(t);      // row_ror:2
    t = add_dpp<0x121>(t);      // row_ror:1
    return t;
}
__device__ __forceinline__ double sum_matrix(double t)
{
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    const v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, t, z, 0, 0, 0);
    double s = d[0];
    s = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, s, 0.0, 0, 0, 0);
    s = add_dpp<0x4E>(s);       // quad_perm:[2,3,0,1]
    s = add_dpp<0xB1>(s);       // quad_perm:[1,0,3,2]
    return s;
}

template <int MODE>
__global__ void chain(double *out, const double *in, int n)
{
    const int lane = threadIdx.x & 63;
    double t = in[lane];
    double total = 0.0;
    for (int it = 0; it < n; ++it) {
        total = MODE ? sum_matrix(t) : sum_butterfly(t);
        t = __builtin_fma(total, 1e-300, in[lane]);        // (the next sum waits for this one)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = total;
}

template <int MODE>
double run(double *d_out, const double *d_in, const char *name, std::vector<double> &got)
{
    const int n = 200000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((chain<MODE>), dim3(256), dim3(256), 0, 0, d_out, d_in, n / 4);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((chain<MODE>), dim3(256), dim3(256), 0, 0, d_out, d_in, n);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    got.resize(64);
    (void)hipMemcpy(got.data(), d_out, 64 * sizeof(double), hipMemcpyDeviceToHost);
    printf("  %-10s %7.1f ns per sum + one multiply-add (one wave per SIMD, every CU)\n", name,
           ms * 1e6 / n);
    return ms * 1e6 / n;
}

int main()
{
    std::vector<double> h(64);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < 64; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = (double)(s >> 11) / 9007199254740992.0 * 1e3 + 1e-3 * i;
    }
    double *d_in, *d_out;
    (void)hipMalloc(&d_in, 64 * sizeof(double));
    (void)hipMalloc(&d_out, 256 * 256 * sizeof(double));
    (void)hipMemcpy(d_in, h.data(), 64 * sizeof(double), hipMemcpyHostToDevice);
    std::vector<double> gb, gm;
    run<0>(d_out, d_in, "butterfly", gb);
    run<1>(d_out, d_in, "matrix", gm);
    // the orders, on the host (volatile: no contraction, no reassociation)
    auto bits = [](double x) { unsigned long long u; memcpy(&u, &x, 8); return u; };
    volatile double c[16], q[4];
    for (int j = 0; j < 16; ++j) {
        volatile double a = 0.0 + h[j];
        a = a + h[16 + j];
        a = a + h[32 + j];
        a = a + h[48 + j];
        c[j] = a;
    }
    for (int j = 0; j < 4; ++j) {
        volatile double a = 0.0 + c[j];
        a = a + c[j + 4];
        a = a + c[j + 8];
        a = a + c[j + 12];
        q[j] = a;
    }
    // quad_perm [2,3,0,1]: lane j adds lane j ^ 2; then [1,0,3,2]: j ^ 1
    volatile double p0 = q[0] + q[2], p1 = q[1] + q[3];
    volatile double tot = p0 + p1;
    printf("  matrix variant, lane 0: %.17g (%016llx); host in the claimed order: %.17g (%016llx): %s\n",
           gm[0], bits(gm[0]), (double)tot, bits((double)tot),
           bits(gm[0]) == bits((double)tot) ? "IDENTICAL" : "different");
    bool same = true;
    for (int l = 1; l < 64; ++l)
        same = same && bits(gm[l]) == bits(gm[0]);
    printf("  every lane holds the same total: %s;  butterfly total %.17g\n", same ? "yes" : "NO", gb[0]);
    return 0;
}
