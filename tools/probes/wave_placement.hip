// wave_placement.hip -- where do the waves of a small workgroup land?  HW_ID of every wave of a
// grid of 4-wave (and 2-wave) workgroups: SIMD, CU, SE, XCC.  (Helper waves that are to share
// one chain's arithmetic only help if they sit on different SIMDs.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/wave_placement tools/probes/wave_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void where(unsigned *out)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
    }
    // stay a while so that the whole grid is resident at once
    for (int i = 0; i < 2000; ++i)
        __builtin_amdgcn_s_sleep(10);
}

int main()
{
    for (int waves : {4, 2, 8}) {
        const int nwg = 256;
        unsigned *d = nullptr;
        hipMalloc(&d, sizeof(unsigned) * 2 * nwg * waves);
        hipLaunchKernelGGL(where, dim3(nwg), dim3(64 * waves), 0, 0, d);
        std::vector<unsigned> h(2 * nwg * waves);
        hipMemcpy(h.data(), d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
        int same_simd = 0, distinct = 0;
        for (int g = 0; g < nwg; ++g) {
            unsigned mask = 0;
            for (int w = 0; w < waves; ++w)
                mask |= 1u << ((h[2 * (g * waves + w)] >> 4) & 3);
            const int n = __builtin_popcount(mask);
            distinct += n;
            same_simd += (n == 1);
        }
        printf("%d waves per workgroup, %d workgroups: %.2f distinct SIMDs per workgroup on "
               "average, %d workgroups with all waves on one SIMD\n", waves, nwg,
               (double)distinct / nwg, same_simd);
        for (int g = 0; g < 3; ++g) {
            printf("  workgroup %d:", g);
            for (int w = 0; w < waves; ++w) {
                const unsigned hw = h[2 * (g * waves + w)];
                printf(" [simd %u cu %u se %u xcc %u]", (hw >> 4) & 3, (hw >> 8) & 15,
                       (hw >> 13) & 7, h[2 * (g * waves + w) + 1] & 15);
            }
            printf("\n");
        }
        hipFree(d);
    }
    return 0;
}
