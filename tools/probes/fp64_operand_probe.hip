// Probe: what does a vector FP64 instruction cost the SIMD by WHERE its operands come from?
// profiles/r02_mfma_f64_issue_probe.txt: 32 independent v_fma_f64 with three vector-register
// sources issue at 5.4-6.2 cycles each, not the nominal 4 (58.6 of 78.6 TFLOP/s) -- the ceiling
// every FP64-issue fraction in DESIGN.md is quoted against.  Is it the register file's read
// ports?  NV independent chains per wave, 1 / 2 / 3 waves per SIMD:
//   vvv  v_fma_f64  x, a(v), x, b(v)        three vector sources
//   svv  v_fma_f64  x, a(s), x, b(v)        one scalar, two vector
//   fmac v_fmac_f64 x, a(s), y(v)           scalar x vector + the destination
//   add  v_add_f64  x, x, b(v)
//   adds v_add_f64  x, a(s), x
//   mul  v_mul_f64  x, a(s), x
//   hipcc --offload-arch=gfx950 -O2 -o fp64_operand_probe fp64_operand_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int NV>
__global__ void chains(double *out, int n, double as_in)
{
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-9, b = 1e-9 * lane, y = 1e-12 * lane;
    const double as = __builtin_bit_cast(
        double, ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(
                     (int)(__builtin_bit_cast(unsigned long long, as_in) >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane(
                        (int)__builtin_bit_cast(unsigned long long, as_in)));
    double x[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
        x[i] = lane * 1e-9 * (i + 1);
    // (UNROLL passes through the chains per trip of the loop: with one instruction per trip the
    // loop's own branch -- a taken branch stalls the wave for several ns -- was half of what the
    // first version of this probe reported for a single chain)
    constexpr int UNROLL = NV >= 16 ? 2 : 32 / NV;
    for (int it = 0; it < n; it += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if constexpr (MODE == 0)
                asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            else if constexpr (MODE == 1)
                asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(x[i]) : "s"(as), "v"(b));
            else if constexpr (MODE == 2)
                asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x[i]) : "s"(as), "v"(y));
            else if constexpr (MODE == 3)
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            else if constexpr (MODE == 4)
                asm volatile("v_add_f64 %0, %1, %0" : "+v"(x[i]) : "s"(as));
            else if constexpr (MODE == 5)
                asm volatile("v_mul_f64 %0, %1, %0" : "+v"(x[i]) : "s"(as));
            else {
                // an addition, then its result through two 32-bit moves (what a level of the
                // wave sum does between two additions: profiles/r06_wave_sum_probe.txt)
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                int lo = __double2loint(x[i]), hi = __double2hiint(x[i]);
                asm volatile("v_mov_b32 %0, %0" : "+v"(lo));
                asm volatile("v_mov_b32 %0, %0" : "+v"(hi));
                x[i] = __hiloint2double(hi, lo);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV>
void run(double *d, int waves, const char *name)
{
    const int n = 61440;     // (tens of milliseconds per launch: the clock has come up)
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // one workgroup per CU, 4 x waves waves: `waves` per SIMD
    hipLaunchKernelGGL((chains<MODE, NV>), dim3(256), dim3(256 * waves), 0, 0, d, n / 4, 1.0000001);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((chains<MODE, NV>), dim3(256), dim3(256 * waves), 0, 0, d, n, 1.0000001);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_op_simd = ms * 1e6 / n / NV / waves;
    printf("  %-8s %2d chains, %d wave(s)/SIMD: %6.2f ns per instruction and SIMD = %5.2f cycles at 2.4 GHz\n",
           name, NV, waves, ns_per_op_simd, ns_per_op_simd * 2.4);
}

template <int NV>
void all(double *d, int waves)
{
    run<0, NV>(d, waves, "vvv");
    run<1, NV>(d, waves, "svv");
    run<2, NV>(d, waves, "fmac");
    run<3, NV>(d, waves, "add");
    run<4, NV>(d, waves, "adds");
    run<5, NV>(d, waves, "mul");
    run<6, NV>(d, waves, "add+2mov");
}

int main(int argc, char **)
{
    const bool quick = argc > 1;          // (any argument: one wave per SIMD only)
    double *d;
    hipMalloc(&d, 64 << 20);
    for (int waves = 1; waves <= 3; ++waves) {
        all<1>(d, waves);
        all<2>(d, waves);
        all<4>(d, waves);
        all<8>(d, waves);
        all<32>(d, waves);
        if (quick)
            break;
    }
    return 0;
}
