// Probe: what does a block of the resident tiled plan's products cost on the FP64 pipe of a SIMD
// when nothing else is in the way?  Eight waves per workgroup (two per SIMD), one workgroup per
// CU; a wave multiplies blocks of sixteen steps -- per step a chain of three dependent
// v_mfma_f64_16x16x4_f64 and four fused multiply-adds on the product, then per block the
// sixteen-lane row totals (16 DPP additions) -- exactly resident.h's instruction mix, with no
// hand-off, no proposal rows, no sums to publish.  PIPE: the products of step s + 1 issued
// before the multiply-adds of step s (resident_async_kernel's order); otherwise the compiler's.
// LDSB: the third operand block read from LDS (two blocks in registers), else all in registers.
// PIPE 2 / 3: the products two steps ahead / two steps' dependent products side by side -- no
// faster than one step ahead (one wave alone: 1.89-1.92 us per block either way; the four
// multiply-adds of a step cost the pipe ~46 cycles beside the matrix instructions' 192).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tile_products_probe tile_products_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ double add_ror(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int l2 = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, false);
    const int h2 = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, false);
    return v + __hiloint2double(h2, l2);
}

template <int PIPE, bool LDSB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void blocks(double *out, unsigned long long *cyc,
                                                     int nblocks)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    constexpr int KR = LDSB ? 2 : 3;
    double bop[16][KR];
#pragma unroll
    for (int st = 0; st < 16; ++st) {
#pragma unroll
        for (int k = 0; k < KR; ++k)
            bop[st][k] = 1.0 + 1e-3 * lane + 1e-5 * st + k;
        if (LDSB)
            lds[st * 64 * WAVES + threadIdx.x] = 0.5 + 1e-4 * st;
    }
    __syncthreads();
    double a[3] = {1.0 + lane * 1e-4, 0.5 - lane * 1e-4, -1.0};
    double total = 0.0;
    int wo = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < nblocks; ++b) {
        asm volatile("; block" : "+v"(wo), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
        double acc[4] = {0, 0, 0, 0};
        auto chain = [&](int st) {
            v4d m = {0, 0, 0, 0};
            m = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], bop[st][0], m, 0, 0, 0);
            m = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], bop[st][1], m, 0, 0, 0);
            m = __builtin_amdgcn_mfma_f64_16x16x4f64(
                a[2], LDSB ? lds[wo + st * 64 * WAVES + threadIdx.x] : bop[st][KR - 1], m, 0, 0, 0);
            return m;
        };
        if (PIPE == 3) {
            // two steps' chains side by side (the dependent products of one between those of
            // the other), a pair of steps ahead of the multiply-adds: four product tuples
            v4d m[4];
            m[0] = chain(0);
            m[1] = chain(1);
#pragma unroll
            for (int st = 0; st < 16; st += 2) {
                if (st + 2 < 16) {
                    m[(st + 2) & 3] = chain(st + 2);
                    m[(st + 3) & 3] = chain(st + 3);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[r] = __builtin_fma(m[st & 3][r], m[st & 3][r], acc[r]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[r] = __builtin_fma(m[(st + 1) & 3][r], m[(st + 1) & 3][r], acc[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (PIPE == 2) {
            // the products TWO steps ahead of the multiply-adds (three product tuples)
            v4d m[3];
            m[0] = chain(0);
            m[1] = chain(1);
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                if (st + 2 < 16)
                    m[(st + 2) % 3] = chain(st + 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[r] = __builtin_fma(m[st % 3][r], m[st % 3][r], acc[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (PIPE) {
            v4d m[2];
            m[0] = chain(0);
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                if (st + 1 < 16)
                    m[(st + 1) & 1] = chain(st + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[r] = __builtin_fma(m[st & 1][r], m[st & 1][r], acc[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const v4d m = chain(st);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[r] = __builtin_fma(m[r], m[r], acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double sg = add_ror<8>(acc[r]);
            sg = add_ror<4>(sg);
            sg = add_ror<2>(sg);
            total += add_ror<1>(sg);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = total;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        cyc[0] = t1 - t0;
}

template <int PIPE, bool LDSB, int WAVES>
void run(double *d, unsigned long long *c, const char *what, int grid = 245)
{
    const int nb = 2000;
    const size_t smem = LDSB ? 16 * 64 * WAVES * sizeof(double) : 0;
    hipFuncSetAttribute(reinterpret_cast<const void *>(blocks<PIPE, LDSB, WAVES>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((blocks<PIPE, LDSB, WAVES>), dim3(grid), dim3(64 * WAVES), smem, 0, d, c, 200);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((blocks<PIPE, LDSB, WAVES>), dim3(grid), dim3(64 * WAVES), smem, 0, d, c, nb);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cy = 0;
    hipMemcpy(&cy, c, sizeof(cy), hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on this part: wall time is the reference
    printf("%3d CUs: %-44s %d waves/SIMD: %7.3f us per block = %6.0f cycles at 2.4 GHz (%5.0f ticks of "
           "s_memtime); matrix instructions alone would take %d x 64 = %d cycles\n",
           grid, what, WAVES / 4, ms * 1e3 / nb, ms * 1e3 / nb * 2400.0, (double)cy / nb,
           48 * WAVES / 4, 48 * 64 * WAVES / 4);
}

int main()
{
    double *d;
    unsigned long long *c;
    hipMalloc(&d, 245 * 1024 * sizeof(double));
    hipMalloc(&c, 64);
    run<0, false, 8>(d, c, "compiler's order, three blocks in registers");
    run<1, false, 8>(d, c, "products a step ahead, three in registers");
    run<0, true, 8>(d, c, "compiler's order, third block in LDS");
    run<1, true, 8>(d, c, "products a step ahead, third block in LDS");
    run<0, false, 4>(d, c, "compiler's order, three blocks in registers");
    run<1, false, 4>(d, c, "products a step ahead, three in registers");
    run<1, true, 4>(d, c, "products a step ahead, third block in LDS");
    run<2, false, 8>(d, c, "products two steps ahead, three in registers");
    run<2, false, 4>(d, c, "products two steps ahead, three in registers");
    run<2, true, 8>(d, c, "products two steps ahead, third block in LDS");
    run<3, false, 8>(d, c, "two steps' products side by side, a pair ahead");
    run<3, false, 4>(d, c, "two steps' products side by side, a pair ahead");
    run<3, true, 8>(d, c, "two steps side by side, third block in LDS");
    // is it the pipe or the power?  the same loop on fewer CUs (a wave's time per block would
    // not change if the pipe of its SIMD were all that bounds it)
    for (int grid : {8, 32, 64, 128, 192, 245, 256})
        run<1, false, 8>(d, c, "products a step ahead, three in registers", grid);
    for (int grid : {8, 64, 245})
        run<1, false, 4>(d, c, "products a step ahead, three in registers", grid);
    return 0;
}
